#!/bin/bash
# One-node launcher for the clip-parallel bench / clip runs: one process per GPU over RCCL (torch.distributed backend "nccl"),
# each rank pinned to ITS SLICE of the CPU cores (and to the memory) of the NUMA node its GPU hangs off, host thread pools capped so that 8 ranks
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
# CPU list and NUMA node of this rank: the node of HIP device $R (PCI address from the HIP runtime -> /sys/bus/pci/devices/<addr>/
# numa_node), its cores SLICED between the ranks whose GPUs share the node (tools/rank_affinity.py); even split without NUMA info
read -r CPUS NODE <<<"$(python "$ROOT/tools/rank_affinity.py" "$R" "$DEMFI_NODE_N" 2>/dev/null || echo "")"
PER=$((NCPU / DEMFI_NODE_N)); [ "$PER" -lt 1 ] && PER=1
if [ -n "$CPUS" ]; then PER=$(echo "$CPUS" | tr ',' '\n' | wc -l); fi
BIND=""
if [ -n "$CPUS" ] && [ "${NODE:--1}" -ge 0 ] && command -v numactl >/dev/null 2>&1; then
  BIND="numactl --physcpubind=$CPUS --membind=$NODE"
elif [ -n "$CPUS" ] && command -v taskset >/dev/null 2>&1; then
  BIND="taskset -c $CPUS"
fi
# host threads: the frame pool (decode / encode) gets this rank's share of the cores, the math libraries stay single-threaded
export DEMFI_IO_THREADS=${DEMFI_IO_THREADS:-$PER} OMP_NUM_THREADS=1 MKL_NUM_THREADS=1
CMD=${DEMFI_NODE_CMD:-"python $ROOT/bench.py --gpus $DEMFI_NODE_N"}
[ "$R" = "0" ] && echo "[run_node] $DEMFI_NODE_N ranks, $NCPU cpus, $PER io threads per rank, rank 0 bound with: ${BIND:-none}" >&2
exec $BIND $CMD $DEMFI_NODE_ARGS
