// fp64 instantiation of the pipeline: the parity build.  Compiled with -ffp-contract=off so that
// every product and sum rounds on its own, as the reference's JavaScript arithmetic does.
#include "rt_pipeline_impl.h"
RT_INSTANTIATE_PIPELINE(double)
