set -e
mkdir -p /tmp/p2ps
for cfg in "8 8192 1500" "4 16384 600" "2 65536 100" "8 65536 60"; do
  set -- $cfg; W=$1; N=$2; S=$3
  rm -f /tmp/p2ps/*; pids=""
  for r in $(seq 0 $((W-1))); do python tests/shard_gpu_worker.py $r $W 29733 $N $S /tmp/p2ps p2p > /tmp/p2ps/log_$r.txt 2>&1 & pids="$pids $!"; done
  ok=1; for p in $pids; do wait $p || ok=0; done
  echo "world=$W n=$N steps=$S ok=$ok"; [ $ok = 1 ] || tail -5 /tmp/p2ps/log_*.txt
done
