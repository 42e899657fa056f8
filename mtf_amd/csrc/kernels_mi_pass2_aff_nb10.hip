/* pass 2 of the MI recompute iteration, affine, up to ten bins (see mtfhip_mi_pass2_unit.h) */
#define MTFHIP_P2_SSM MTFHIP_SSM_AFFINE
#define MTFHIP_P2_MC false
#define MTFHIP_P2_NB 10
#define MTFHIP_P2_NAME launch_mi_pass2_aff_nb10
#include "mtfhip_mi_pass2_unit.h"
