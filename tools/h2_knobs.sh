#!/bin/bash
# GPU box: the h2 renderer's kernel time per 1,024 rays x (64+128) under compile-time knobs (rebuilds the library per variant).
for fl in "" "-DCRNERF_H2_AHEAD=4" "-DCRNERF_H2_AHEAD=16" "$@"; do
  CRNERF_EXTRA_FLAGS="$fl" python cr-nerf-pytorch_amd/build.py > /dev/null 2>&1 || { echo "[$fl] build failed"; continue; }
  echo "[$fl] $(CRNERF_EXTRA_FLAGS="$fl" python bench.py --precision f32h2 --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c 'import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print("kernel_ms %.4f  step %.4f ms" % (j["roofline"]["kernel_ms"], j["ms_per_step"]))')"
done
