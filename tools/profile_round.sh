#!/bin/bash
# Regenerates the per-round profile set under gpurun_out/<tag>/ on the GPU box (copy what is to be judged into profiles/):
#   bash tools/profile_round.sh r02          (run through gpurun; ~6 minutes of box time)
# Counter passes are separate rocprofv3 runs with --pmc only (no trace domains), the program itself after "--".
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --host-input 0 --kernel-roofline 0 --other-configs 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --graph 0 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --graph 0 > $OUT/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json > $OUT/pmc_traffic.log 2>&1 || exit 1
# the other BASELINE configurations' traffic (bench.py other_configs reports it beside their img/s)
for cfg in "--backbone densenet:densenet_S7:densenet S=7 batch 64" "--S 14:resnet_S14:resnet S=14 batch 64"; do
  flags=${cfg%%:*}; rest=${cfg#*:}; name=${rest%%:*}; wl=${rest#*:}
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$name -- $BENCH $flags --graph 0 > $OUT/pmc_fetch_$name.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$name -- $BENCH $flags --graph 0 > $OUT/pmc_write_$name.log 2>&1 || exit 1
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$name $OUT/pmc_write_$name $OUT/${TAG}_${name}_pmc_traffic.json "$wl" >> $OUT/pmc_traffic.log 2>&1 || exit 1
  rm -rf $OUT/pmc_fetch_$name $OUT/pmc_write_$name
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_sq -- $BENCH --graph 0 > $OUT/pmc_sq.log 2>&1 || exit 1
python3 tools/pmc_sq.py $OUT/pmc_sq > $OUT/${TAG}_pmc_sq_summary.txt 2>&1 || exit 1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --host-input 0 --kernel-roofline 0 --other-configs 0 > $OUT/prof.log 2>&1 || exit 1
python3 tools/prof_summary.py $OUT/prof > $OUT/${TAG}_bench_kernel_summary_last_step.txt 2>&1 || exit 1
python3 tools/trace_gaps.py $OUT/prof >> $OUT/${TAG}_bench_kernel_summary_last_step.txt 2>&1 || exit 1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_rocprofv3_kernel_stats.csv
rm -rf $OUT/prof
for cfg in "--backbone densenet:densenet121_S7" "--S 14:resnet50_S14" "--S 14 --fp8-forward:resnet50_S14_fp8_forward"; do
  flags=${cfg%%:*}; name=${cfg##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python3 bench.py $flags --steps 3 --warmup 2 --no-cpu-baseline --host-input 0 --kernel-roofline 0 --other-configs 0 > $OUT/prof_$name.log 2>&1 || exit 1
  python3 tools/prof_summary.py $OUT/prof_$name > $OUT/${TAG}_${name}_kernel_summary_last_step.txt 2>&1 || exit 1
  rm -rf $OUT/prof_$name
done
python3 tools/bench_conv.py > $OUT/${TAG}_conv_layer_table.txt 2>&1 || exit 1
python3 tools/bench_bn.py > $OUT/${TAG}_bn_kernel_table.txt 2>&1 || exit 1
echo done > $OUT/done.txt
