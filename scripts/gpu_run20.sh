timeout 900 python bench.py --no-cpu-baseline --controlnet 2>&1 | grep -v amdgpu | cut -c1-900
