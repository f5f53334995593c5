// Interface between the two translation units of libmpbp_hip: the batched gauge sweep (v2_engine.hip) and the sweep
// driver / workgroup-per-problem engine (mpbp_hip.hip).
#pragma once
#include "ctx.h"
