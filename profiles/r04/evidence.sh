# Round-4 evidence on ONE box: bench lines of every BASELINE config + variants, rocprofv3 kernel traces, PMC passes (incl. FETCH / WRITE_SIZE).
#   gpurun -- 'bash profiles/r04/evidence.sh r04final'   ->  gpurun_out/<tag>/ ; what is cited is copied into profiles/r04/
T=${1:-r04final}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/$T && cd gpurun_out/$T
python $R/bench.py > bench_config2.json 2> c2.err
python $R/bench.py --steps 20 --warmup 5 > bench_driver_cmd_20_5.json 2>> c2.err
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > bench_driver_cmd_20_5_b.json 2>> c2.err
python $R/bench.py --sequences 1 --no-cpu-baseline > bench_config2_one_sequence.json 2>> c2.err
python $R/bench.py --resident -1 --no-cpu-baseline > bench_config2_launch_per_bounce.json 2>> c2.err
python $R/bench.py --rotat degrees --no-cpu-baseline > bench_config2_degrees.json 2>> c2.err
python $R/bench.py --direct-light --no-cpu-baseline > bench_config2_direct_light.json 2>> c2.err
python $R/bench.py --config 1 --steps 256 > bench_config1.json 2> c1.err
python $R/bench.py --config 3 > bench_config3.json 2> c3.err
python $R/bench.py --config 4 --steps 128 > bench_config4_n1.json 2> c4.err
python $R/bench.py --config 5 --steps 512 > bench_config5.json 2> c5.err
python $R/bench.py --config 5 --steps 512 --resident -1 --no-cpu-baseline > bench_config5_launch_per_bounce.json 2>> c5.err
echo benches done
cd $R
python3 profiles/collect_pmc.py --tag ${T}_c2 > gpurun_out/$T/pmc_c2.log 2>&1
python3 profiles/collect_pmc.py --tag ${T}_c3 -- --config 3 > gpurun_out/$T/pmc_c3.log 2>&1
python3 profiles/collect_pmc.py --tag ${T}_c5 -- --config 5 > gpurun_out/$T/pmc_c5.log 2>&1
python3 profiles/collect_pmc.py --tag ${T}_c2_seq1 --passes trace -- --sequences 1 > gpurun_out/$T/trace_c2_seq1.log 2>&1
for f in gpurun_out/$T/bench_*.json; do python -c "
import sys,json
p=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], round(p['value']), p.get('value_cold') and round(p['value_cold']), round(p['roofline']['frac'],3), round(p['roofline']['kernel_alone']['frac'],3), p.get('cpu_baseline',{}).get('value'))
" $f; done
for t in c2 c3 c5 c2_seq1; do python -c "
import sys,json
s=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-2], json.dumps(s.get('k_bounce_trace_union')), json.dumps({k:v for k,v in s.get('k_bounce_derived',{}).items() if k in ('hbm_bytes_per_iteration','valu_active_lanes_avg','trace_avg_launch_ms','executed_fp32_flops_per_ray_bounce','SQ_LDS_BANK_CONFLICT','SQ_ACTIVE_INST_LDS')}))
" gpurun_out/pmc_${T}_$t/summary.json; done
# keep the merge small: raw traces are large
find gpurun_out/pmc_${T}_* -name "*kernel_trace.csv" -size +20M -delete
