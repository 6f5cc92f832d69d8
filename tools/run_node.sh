#!/bin/bash
# One-node launcher for the clip-parallel bench / clip runs: one process per GPU over RCCL (torch.distributed backend "nccl"),
# each rank pinned to the CPU cores and memory of the NUMA node its GPU hangs off, host thread pools capped so that 8 ranks
# do not oversubscribe the host (the PNG codec and the pinned-memory copies are the only host work: SURVEY.md section 8e/8f).
#
#   tools/run_node.sh [N_GPUS] [bench.py arguments ...]          e.g.  tools/run_node.sh 8 --steps 20 --warmup 5
#   DEMFI_NODE_CMD="python my_clip_job.py" tools/run_node.sh 8    (any script that reads RANK / LOCAL_RANK / WORLD_SIZE)
#
# The driver's own launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N) works without
# this wrapper; the wrapper only adds the affinity: it re-executes itself once per rank (DEMFI_NODE_CHILD=1) under torchrun.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ -z "$DEMFI_NODE_CHILD" ]; then
  N=${1:-8}; shift || true
  export DEMFI_NODE_CHILD=1 DEMFI_NODE_ARGS="$*" DEMFI_NODE_N=$N
  export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29533}" \
       --no-python "$0"
fi
# ---- per-rank child -----------------------------------------------------------------------------------------------
R=${LOCAL_RANK:-0}
NCPU=$(nproc)
# NUMA node of GPU R: /sys/class/drm/cardX/device/numa_node of the R-th render-capable amdgpu device (falls back to an even split)
NODE=-1
i=0
for d in /sys/class/drm/card*/device; do
  [ -e "$d/vendor" ] && [ "$(cat "$d/vendor")" = "0x1002" ] || continue
  if [ "$i" = "$R" ]; then NODE=$(cat "$d/numa_node" 2>/dev/null || echo -1); break; fi
  i=$((i + 1))
done
PER=$((NCPU / DEMFI_NODE_N)); [ "$PER" -lt 1 ] && PER=1
if [ "$NODE" -ge 0 ] && command -v numactl >/dev/null 2>&1; then
  # cores of that NUMA node, split evenly between the ranks that share it
  BIND="numactl --cpunodebind=$NODE --membind=$NODE"
else
  LO=$((R * PER)); HI=$((LO + PER - 1))
  BIND="taskset -c $LO-$HI"
  command -v taskset >/dev/null 2>&1 || BIND=""
fi
# host threads: the frame pool (decode / encode) gets this rank's share of the cores, the math libraries stay single-threaded
export DEMFI_IO_THREADS=${DEMFI_IO_THREADS:-$PER} OMP_NUM_THREADS=1 MKL_NUM_THREADS=1
CMD=${DEMFI_NODE_CMD:-"python $ROOT/bench.py --gpus $DEMFI_NODE_N"}
[ "$R" = "0" ] && echo "[run_node] $DEMFI_NODE_N ranks, $NCPU cpus, $PER io threads per rank, rank 0 bound with: ${BIND:-none}" >&2
exec $BIND $CMD $DEMFI_NODE_ARGS
