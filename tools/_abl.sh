export DEMFI_HIP_LIB=$PWD/demfi_amd/csrc/libdemfi_hip_abl.so
for data in zero relu; do
for v in 0 1 3 15 2 4 7 9; do
  echo "data=$data var=$v: $(PROBE_DATA=$data DEMFI_PERSIST_VARIANT=$v python tools/conv_probe.py c3x3 40 2>&1 | grep -v amdgpu.ids | tail -1)"
done; done
