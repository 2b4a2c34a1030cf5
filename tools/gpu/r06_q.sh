#!/usr/bin/env bash
# round 6, call Q: PMC occupancy evidence of the saturated launch (N = 65536, 8192, 1024): SQ_WAVE_CYCLES / SQ_BUSY_CYCLES per SE ~ resident
# waves, VALU instructions per wave, wait fractions (separate --pmc run, kernel trace only)
set -u
R=$PWD
OUT=$R/gpurun_out/r06_q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1024 8192 65536; do
  JAXSIM_AMD_SPECIALIZE=1 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_$n -- python $R/tools/sweep.py --sizes $n --steps 60 --reps 2 > $OUT/pmc_$n.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for n in (1024, 8192, 65536):
    files = glob.glob(f'gpurun_out/r06_q/pmc_{n}/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if 'jxs_kernel' in row.get('Kernel_Name', ''):
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
    if not acc:
        print(n, 'no counters'); continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    waves = m.get('SQ_WAVES', 0)
    print(f"N={n}: launches {len(acc['SQ_WAVES'])}, waves/launch {waves:.0f}, SQ_WAVE_CYCLES/SQ_BUSY_CYCLES {m.get('SQ_WAVE_CYCLES',0)/max(1,m.get('SQ_BUSY_CYCLES',1)):.2f}, "
          f"VALU/wave {m.get('SQ_INSTS_VALU',0)/max(1,waves):.0f}, WAIT_ANY/WAVE_CYCLES {m.get('SQ_WAIT_ANY',0)/max(1,m.get('SQ_WAVE_CYCLES',1)):.2f}, "
          f"WAIT_INST_ANY/WAVE_CYCLES {m.get('SQ_WAIT_INST_ANY',0)/max(1,m.get('SQ_WAVE_CYCLES',1)):.2f}, wave cycles per wave {m.get('SQ_WAVE_CYCLES',0)/max(1,waves)*4:.0f} (x4: quad-cycles), GRBM_GUI_ACTIVE {m.get('GRBM_GUI_ACTIVE',0):.0f}")
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
