# A/B of the speculative loop search's pool size (LSLAM_FE_LOOP_POOL) on config 5 at full size
for n in 1 2 4 6 8 12; do
  echo -n "LSLAM_FE_LOOP_POOL=$n  "
  LSLAM_FE_LOOP_POOL=$n timeout 300 python tools/bench_extra.py --only cfg5 --stream-ref 0 2>/dev/null | grep "^{" | python -c "
import sys,json
j=json.loads(sys.stdin.readline()); print(j['gpu_scans_per_s'], j['graph']['edges'], j['graph']['loop_coarse_matches'], j['graph']['loops_closed'], j['map_bit_exact'])"
done
