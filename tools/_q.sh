timeout 900 python -m pytest tests -m gpu -x -q -k "surface or crowded or radius or overflow or normal" 2>&1 | tail -2
timeout 300 python tools/fuzz_parity.py 150 5 2>&1 | tail -1
timeout 300 python tools/time_surface.py 2>&1 | grep "surface"
timeout 300 python tools/time_small.py 2>&1 | grep "surface" | grep "auto"
