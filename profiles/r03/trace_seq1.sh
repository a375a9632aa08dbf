T=${1:-r03r}
python3 profiles/collect_pmc.py --tag ${T}_c2_seq1 --passes trace -- --sequences 1 > /dev/null 2>&1
python3 profiles/collect_pmc.py --tag ${T}_c2_seq2 --passes trace > /dev/null 2>&1
python3 profiles/collect_pmc.py --tag ${T}_c5_seq1 --passes trace -- --sequences 1 --config 5 > /dev/null 2>&1
for t in c2_seq1 c2_seq2 c5_seq1; do python -c "
import sys,json
s=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-2], {k:(v.get('calls'),round(v.get('avg_us',0),1),round(v.get('pct',0),1)) for k,v in s['kernels'].items()}, s.get('k_bounce_trace_union',{}).get('eff_ms'), s.get('bench_under_trace',{}).get('value'))
" gpurun_out/pmc_${T}_$t/summary.json; done
grep -h "k_accumulate\|k_iter" gpurun_out/pmc_${T}_c2_seq1/trace/trace_kernel_stats.csv | cut -c1-150
