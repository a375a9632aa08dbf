# diagnostic builds of the pair kernel on config 2: pairs per ray by type, batches per round (PT_DEBUG_PAIR), clocks per phase of a later-bounce round (PT_DEBUG_PHASE=2)
O=gpurun_out/$1; mkdir -p $O
PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_pairdbg/libptamd.so PT_DEBUG_PAIR=1 python bench.py --no-cpu-baseline --steps 32 --sequences 1 > $O/pair.json 2> $O/pair.err
PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_phase2/libptamd.so PT_DEBUG_PHASE2=1 python bench.py --no-cpu-baseline --steps 32 --sequences 1 > $O/phase.json 2> $O/phase.err
grep -h "ptamd" $O/pair.err $O/phase.err | sort | uniq -c | sort -rn | head -12
