cd $GRAFT_REPO_ROOT
for b in 64 96 128 192 256; do for r in 2 3; do echo "budget $b rmax $r: $(OA_GRID_BUDGET=$b OA_GRID_RMAX=$r timeout 120 python tools/cold_surface.py 4 5 2>&1 | tail -1)"; done; done
for cell in 1.25 1.5 2.0; do echo "tri cell $cell: $(OA_TRI_CELL=$cell timeout 120 python tools/cold_surface.py 4 5 2>&1 | tail -1)"; done
