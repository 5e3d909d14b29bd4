// fdh_composite_uniform.hip -- the second translation unit of fdh_kernels.hip: k_composite_tiles<0|2|4> (every build but
// the one with the one-pixel-slot path) and their launcher, compiled with -structurizecfg-skip-uniform-regions.
// Why, and why only these kernels: the FDH_TU note at the top of fdh_kernels.hip.
#define FDH_TU 1
#include "fdh_kernels.hip"
