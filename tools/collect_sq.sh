#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:   bash tools/collect_sq.sh <tag> [prof_target args...]
# SQ counter passes (8 per pass, --kernel-trace only) of tools/prof_target.py (or tools/$GX_PROF_SCRIPT); tools/summarize_sq.py turns the CSVs
# into one small table under gpurun_out/<tag>/sq.txt.
set -eo pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
p=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH"; do
  p=$((p+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/sq$p" -o sq -- python3 "$root/tools/${GX_PROF_SCRIPT:-prof_target.py}" "$@" > /dev/null 2> "$out/sq$p.err" || { echo "pass $p failed"; tail -3 "$out/sq$p.err"; }
  echo "pass $p done"
done
cd "$root"
python3 tools/summarize_sq.py "$out" > "$out/sq.txt"
find "$out" -type f ! -name "sq.txt" ! -name "*.err" -delete
cat "$out/sq.txt"
