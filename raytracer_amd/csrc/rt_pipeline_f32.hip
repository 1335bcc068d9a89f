// fp32 instantiation of the pipeline: the throughput build (FMA contraction on).
#include "rt_pipeline_impl.h"
RT_INSTANTIATE_PIPELINE(float)
