// fdh_kernels_all.hip -- every kernel family in ONE translation unit, for the instrumented builds (`make stats`, `make variant SINGLE=1`:
// FDH_STATS / FDH_TIMING keep device-side counters, which must be one set).  The product build compiles the units one by one (Makefile).
#include "k_bin_upload.hip"
#include "k_composite.hip"
#include "k_blur_valu.hip"
#include "k_blur_mx.hip"
#include "k_atlas_upload.hip"
