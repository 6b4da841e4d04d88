#!/bin/bash
python -m pytest tests/test_gpu_sym.py -m gpu -q 2>&1 | tail -1
for rep in 1 2 3; do python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sym: value %.4e ms/step %.4f kernel %.4f clk %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['held_clock_ghz']))"; done
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --kernel sgpr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one-sided: value %.4e ms/step %.4f' % (d['value'], d['ms_per_step']))"
