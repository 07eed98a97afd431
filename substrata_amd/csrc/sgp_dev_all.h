// sgp_dev_all.h -- every shared device header of the step kernels, in dependency order.
#pragma once
#include "sgp_dev_common.h"
#include "sgp_dev_broadphase.h"
#include "sgp_dev_narrowphase.h"
#include "sgp_dev_meshpair.h"
#include "sgp_dev_constraints.h"
#include "sgp_dev_solve.h"
#include "sgp_dev_sweep.h"
#include "sgp_dev_edits.h"
#include "sgp_dev_queries.h"
#include "sgp_dev_vehiclecast.h"
