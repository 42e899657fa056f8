/* pass 2 of the MI recompute iteration, affine (see mtfhip_mi_pass2_unit.h) */
#define MTFHIP_P2_SSM MTFHIP_SSM_AFFINE
#define MTFHIP_P2_MC false
#define MTFHIP_P2_NAME launch_mi_pass2_aff
#include "mtfhip_mi_pass2_unit.h"
