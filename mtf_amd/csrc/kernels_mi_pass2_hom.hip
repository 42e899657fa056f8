/* pass 2 of the MI recompute iteration, homography (see mtfhip_mi_pass2_unit.h) */
#define MTFHIP_P2_SSM MTFHIP_SSM_HOMOGRAPHY
#define MTFHIP_P2_MC false
#define MTFHIP_P2_NAME launch_mi_pass2_hom
#include "mtfhip_mi_pass2_unit.h"
