cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('events on ', d['value'], d['ms_per_step'], d['secondary'].get('l2hmc_hip_graph'))"
L2Q_BENCH_NO_KTIMER=1 python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe 2>gpurun_out/kt/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('events off', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/kt/err.txt
done
