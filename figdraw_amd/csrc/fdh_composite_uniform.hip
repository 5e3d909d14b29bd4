// fdh_composite_uniform.hip -- the second translation unit of fdh_kernels.hip: k_composite_tiles<4> (phases without clip
// operations) and its launcher, compiled with -structurizecfg-skip-uniform-regions.  Why, and why only this kernel: the
// FDH_TU note at the top of fdh_kernels.hip.
#define FDH_TU 1
#include "fdh_kernels.hip"
