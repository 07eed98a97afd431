// sgp_kernels_experiments.hip -- the step kernels as ONE translation unit plus the experiments (python -m substrata_amd.build --experiments):
// the solver probe (tools/solve_probe.py) and the resident tile solver of round 3 (SGP_TILE_SOLVER=1, tests/test_tile_solver_gpu.py) reach into several
// stages' kernels, so they are compiled in a unity build instead of the per-stage objects.  Measured negatives and timing aids; the product does not carry them.
#ifndef SGP_EXPERIMENTS
#error "built only with -DSGP_EXPERIMENTS"
#endif
#include "sgp_k_broadphase.hip"
#include "sgp_k_narrowphase.hip"
#include "sgp_k_mesh.hip"
#include "sgp_k_constraints.hip"
#include "sgp_k_solve.hip"
#include "sgp_k_sweep.hip"
#include "sgp_k_edits.hip"
#include "sgp_k_queries.hip"
#include "sgp_k_vehicle.hip"
#include "sgp_k_tiles.hip"
#include "experiments/sgp_solver_probe.inc"
#include "experiments/sgp_tile_solver.inc"
