#!/bin/bash
# GPU box: record budget of the triangle-grid search (queries over budget go to the tree) with the shared scan
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for b in 192 384 768 1536; do for m in 1.0 2.0 4.0; do
  echo "OA_GRID_BUDGET=$b OA_GRID_BUDGET_MOVING=$m: $(OA_GRID_BUDGET=$b OA_GRID_BUDGET_MOVING=$m ONLY=surface:grid python $REPO/tools/time_surface.py 2>&1 | grep iters | sed 's/.*iters/iters/' | tr '\n' ' ')"
done; done
