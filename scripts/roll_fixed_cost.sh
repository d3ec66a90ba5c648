#!/bin/bash
# Where does a rollout launch's time go beyond steps x slope?  (1) launches back to back at several steps-per-launch (structural
# fixed cost: intercept of T(c)); (2) one launch from an idle chip (what --steps K --repeats R times: every repeat follows a
# synchronise); (3) the same after a long untimed warm-up (does the memory/fabric clock state survive the synchronise?)
mkdir -p gpurun_out/r04
O=gpurun_out/r04/roll_fixed.txt
: > $O
line='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%.2f us/step  %.1f us/launch  min/max %s  frac %.3f" % (d["ms_per_step"]*1e3, r["kernel_ms"]*1e3, [round(x*1e3,1) for x in r["kernel_ms_min_max"]], r["frac"]))'
for c in 5 10 20 40 64; do
  echo "== back-to-back, $c steps per launch, 640 steps" >> $O
  TRON_ROLL_CHUNK=$c python bench.py --only-rollout --no-dqn --no-cpu-baseline --steps 640 --warmup 64 --repeats 5 2>/dev/null | python3 -c "$line" >> $O
done
for k in 5 10 20 40 64; do
  echo "== one launch of $k steps after --warmup 5" >> $O
  python bench.py --only-rollout --no-dqn --no-cpu-baseline --steps $k --warmup 5 --repeats 9 2>/dev/null | python3 -c "$line" >> $O
done
for w in 64 640 6400; do
  echo "== one launch of 20 steps after --warmup $w" >> $O
  python bench.py --only-rollout --no-dqn --no-cpu-baseline --steps 20 --warmup $w --repeats 9 2>/dev/null | python3 -c "$line" >> $O
done
cat $O
