# SQ counters of the sliding-window convolution kernels (tools/slide_lab, 32 images 256x256, product variants only), three counter
# sets in three passes; per launch averages.  usage (GPU box): bash tools/pmc_slide.sh > profiles/<tag>_pmc_sq_slide.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_slide -- tools/build/slide_lab_plain 2 1 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob('gpurun_out/pmc_slide/**/*counter_collection.csv', recursive=True))
if not f: print('no counter file'); sys.exit()
rows = list(csv.DictReader(open(f[-1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    if 'conv3x3_slide' not in k: continue
    key = k.split('conv3x3_')[1].split('(')[0]
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    n[(key, r['Counter_Name'])] += 1
for key, d in sorted(agg.items()):
    print(key, ' '.join(f'{c}={v / n[(key, c)]:.4g}' for c, v in d.items()))
PY
  rm -rf gpurun_out/pmc_slide
done
