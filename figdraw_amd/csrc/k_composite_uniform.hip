// k_composite_uniform.hip -- the second translation unit of k_composite.hip: k_composite_tiles<0|2|4> (every build without the
// one-pixel-slot path), k_composite_deep and their launcher, compiled with -structurizecfg-skip-uniform-regions.
// Why, and why only these kernels: the FDH_TU note at the top of k_composite.hip.
#define FDH_TU 1
#include "k_composite.hip"
