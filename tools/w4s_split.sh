cd /root/repo
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/hbm_stream tools/ubench/hbm_stream.hip 2>/dev/null
timeout 120 /tmp/hbm_stream > gpurun_out/hbm_stream.txt 2>&1
export TMPDIR=/tmp
for s in base.17 base.19 base.24 head.0 head.1; do
  rm -rf /tmp/prof_$s
  TILES=44 ITERS=10 timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$s -o p -- python tools/wino_one.py $s > /tmp/prof_$s.log 2>&1
  f=$(find /tmp/prof_$s -name "*kernel_stats.csv" | head -1)
  echo "== $s" >> gpurun_out/w4s_split.txt
  grep -i "wino4s" $f | cut -d, -f1-4 | cut -c1-120 >> gpurun_out/w4s_split.txt
done
