#!/bin/bash
# usage (GPU box, repo root): tools/ab_mesh.sh  -- ms per 500 MD steps of 24 x alanine dipeptide, single-launch mesh pipeline vs three launches
for rep in 1 2; do
for m in 1 0; do echo "REMD_PME_MESH1=$m"; REMD_PME_MESH1=$m timeout 300 python tools/launch_bound_check.py 24 2>&1 | tail -2; done
done
