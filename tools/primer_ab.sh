# bench.py's headline with and without the untimed replay of the timed graph in front of the barrier, alternating processes
for rep in 1 2 3 4 5; do
for P in 1 0; do
  echo -n "CFDP_BENCH_PRIMER=$P: "
  CFDP_BENCH_PRIMER=$P python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-finest --no-loopback --no-irregular --no-power 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['value'],1), round(d['ms_per_step']*1e3,2), 'us/step; kernel', round(d['roofline']['us_per_launch'],2), 'uncond', round(d['config']['unconditioned_ms_per_step']*1e3,2), 'untimed', d['config']['untimed_steps_in_front_of_the_timed_region'])"
done
done
