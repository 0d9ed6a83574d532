#!/bin/bash
# Round profiles of the headline bench (run on the GPU box from the repo root: bash scripts/take_profiles.sh r02):
#  1. rocprofv3 --kernel-trace --stats of the DRIVER's command           -> gpurun_out/<tag>_decode_kernels.md, <tag>_kernel_stats.csv
#  2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes     -> gpurun_out/<tag>_pmc_traffic.{md,json}
#     (never combined with a trace domain; counters corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2 on gfx950)
#  3. gpurun_out/bench_roofline_ref.json: what bench.py reports as roofline.traffic / rocprof_kernel_us (copy to profiles/)
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-config1 --no-config3 --no-config5 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err
T=$(find $OUT/prof_$TAG -name "*kernel_trace.csv" | head -1)
S=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 $ROOT/scripts/summarize_trace.py $T > $OUT/${TAG}_decode_kernels.md
[ -n "$S" ] && head -40 $S > $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o f -- python3 $ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-config1 --no-config3 --no-config5 > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o w -- python3 $ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-config1 --no-config3 --no-config5 > $OUT/${TAG}_pmc_write.log 2>&1
python3 $ROOT/scripts/summarize_pmc.py $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.md
TAG=$TAG OUT=$OUT python3 - <<'PY'
import json, os
tag, out = os.environ["TAG"], os.environ["OUT"]
pmc = json.load(open(f"{out}/{tag}_pmc_traffic.json"))["kernels"]
name = next((k for k in pmc if k.startswith("moe_flat_kernel")), None) or next((k for k in pmc if k.startswith("moe_fused_kernel")), None) \
    or next(k for k in pmc if k.startswith("wstream_gemm<14, 1, 0, 2, 8, true"))
us = None
for line in open(f"{out}/{tag}_decode_kernels.md"):
    if name in line:
        us = float(line.split("|")[3])
json.dump({"kernel": name, "traffic_bytes": pmc[name]["traffic_bytes"], "kernel_us": us,
           "source": f"profiles/{tag}_pmc_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x 2) and profiles/{tag}_decode_kernels.md "
                     f"(rocprofv3 --kernel-trace of `python3 bench.py --gpus 1 --steps 20 --warmup 5`); static: not measured by the bench run itself"},
          open(f"{out}/bench_roofline_ref.json", "w"), indent=1)
print(open(f"{out}/bench_roofline_ref.json").read())
PY
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
head -16 $OUT/${TAG}_decode_kernels.md
