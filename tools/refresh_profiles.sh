#!/bin/bash
# Regenerate the measurement set under profiles/ on a GPU box (one MI355X):  bash tools/refresh_profiles.sh r02a
# Writes scratch under gpurun_out/ and copies the judged summaries to profiles/<tag>_*.
set -e -o pipefail
set -x          # progress on stderr: a GPU box kills a command that stays silent for 7 minutes
tag=${1:?usage: refresh_profiles.sh <tag>}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
python bench.py > gpurun_out/${tag}_bench.log 2>&1 && tail -1 gpurun_out/${tag}_bench.log > profiles/${tag}_bench.json
python bench.py --mode train --prec bf16x3 --steps 10 --warmup 3 2>/dev/null | tail -1 > profiles/${tag}_bench_train.json
python bench.py --mode train --prec f16f4 --steps 10 --warmup 3 2>/dev/null | tail -1 > profiles/${tag}_bench_train_f16f4.json
python bench.py --mode extract --steps 5 --warmup 2 2>/dev/null | tail -1 > profiles/${tag}_bench_extract.json
python bench.py --mode tedm --batch 16 --steps 5 --warmup 2 2>/dev/null | tail -1 > profiles/${tag}_bench_tedm.json
python bench.py --streams 2 --no-extras 2>/dev/null | tail -1 > profiles/${tag}_bench_two_streams.json
python bench.py --mode sample --batch 4 --steps 50 --warmup 3 2>/dev/null | tail -1 > profiles/${tag}_bench_sample.json
python bench.py --no-cpu-baseline --breakdown 2> profiles/${tag}_bench_hip_event_breakdown.txt > /dev/null
# the profiled program comes directly after `--` (no env/bash hop), counters in their own passes
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp gpurun_out/${tag}_stats/*/*_kernel_stats.csv profiles/${tag}_bench_bs32_256_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write profiles/${tag}_pmc_traffic.json
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_train -o train -- python bench.py --mode train --steps 3 --warmup 1 > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/${tag}_train/train_results.db 4 70 > profiles/${tag}_train_kernel_stats.txt
python tools/conv_shape_profile.py > profiles/${tag}_conv_shape_profile.txt 2>/dev/null
# per-dispatch timeline of one forward step (every kernel in launch order)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/trace_one_step.py gpurun_out/${tag}_trace > profiles/${tag}_step_timeline.txt
# sliding-window conv kernels: lab timings with phase stamps (tools/build/slide_lab: hipcc ... -DSL_LAB_EXTRA tools/slide_lab.hip) and SQ counters
if [ -x tools/build/slide_lab ]; then tools/build/slide_lab 10 3 > profiles/${tag}_slide_lab.txt 2>&1; fi
if [ -x tools/build/slide_lab_plain ]; then bash tools/pmc_slide.sh > profiles/${tag}_pmc_sq_slide.txt 2>&1; fi
# (the eps margin scan against the CPU oracle takes ~10 minutes of host time: python tests/eps_margin.py 4 > profiles/${tag}_eps_margin.txt)
# gpurun merges only gpurun_out/ back: leave a copy there (copy it into profiles/ after the call)
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}_* gpurun_out/profiles_${tag}/
echo "profiles/${tag}_* written (copy under gpurun_out/profiles_${tag}/)"
