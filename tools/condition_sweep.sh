for rep in 1 2 3; do
for S in ${SWEEP:-0.25 1.0 0.5 2.0}; do
  echo -n "CONDITION_S=$S: "
  CFDP_BENCH_CONDITION_S=$S python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-finest --no-loopback --no-irregular --no-power 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['value'],1), round(d['ms_per_step']*1e3,2), 'us/step; kernel', round(d['roofline']['us_per_launch'],2), 'uncond', round(d['config']['unconditioned_ms_per_step']*1e3,2), 'untimed', d['config']['untimed_steps_in_front_of_the_timed_region'])"
done
done
