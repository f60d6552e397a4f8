(echo "# python tools/time_surface.py  (1M queries, 980k-vertex / 1.96M-triangle target)"; timeout 300 python tools/time_surface.py 2>&1 | grep -v amdgpu.ids
 echo; echo "# PARTIAL=1 python tools/time_surface.py  (target = z > 0 half: ~47% of the queries have no partner within thresh)"; PARTIAL=1 timeout 300 python tools/time_surface.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r01g_surface.txt
(echo "# python tools/time_small.py"; timeout 300 python tools/time_small.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r01g_small.txt
(echo "# python tools/time_crossover.py"; timeout 600 python tools/time_crossover.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r01g_crossover.txt
