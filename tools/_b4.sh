cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in 2 5; do for B in 4 32; do CTDET_STREAMS=$S python $R/bench.py --steps 30 --warmup 6 --batch $B --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=$S bs=$B', d['value'], d['ms_per_step'], d['launch_mode'])"; done; done
