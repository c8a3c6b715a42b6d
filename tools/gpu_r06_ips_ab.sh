mkdir -p gpurun_out/r06
for r in 1 2; do
for n in ${IPS_LIST:-8 16}; do
  python bench.py --images-per-step $n --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --dedup-steps 0 --no-power 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('ips',$n,'value',round(d['value'],4),'ms_per_step',round(d['ms_per_step'],1))"
done; done | tee gpurun_out/r06/images_per_step_ab${IPS_TAG}.log
