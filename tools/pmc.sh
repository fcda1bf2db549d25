#!/bin/bash
# PMC passes for one bench-like run (each pass in its own process, as the pool requires).
# usage: tools/pmc.sh <outdir> <scene> [accel] [n] [width height [spp_grid]] ; writes CSVs under gpurun_out/<outdir>
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/$1; scene=${2:-cube}; accel=${3:-auto}; n=${4:-20}; shift 4 2>/dev/null || shift $#
size="$*"
mkdir -p $out
run() { # name counters...
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/one_scene.py $scene $accel $n $size > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU || exit 1
# the executed vector instructions by class (wave-instructions)
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 || exit 1
run sq4 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 || exit 1
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run tcc2 WRITE_SIZE GRBM_COUNT || exit 1
python3 tools/pmc_summary.py $out $n
