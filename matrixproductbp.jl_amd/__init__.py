"""MI355X-native MPBP message-update hot path (drop-in for stecrotti/MatrixProductBP.jl's
iterate!/onebpiter! path).  The compute lives in csrc/ (HIP, gfx950) behind the C ABI of
include/mpbp_hip.h; this package is the thin host mirror of the reference's interface."""
from . import _lib
from ._lib import MPBPError, build
from .factors import (BPFactor, DampedFactor, GenericFactor, GenericGlauberFactor, HomogeneousGlauberFactor,
                      IntegerGlauberFactor, PMJGlauberFactor, RecursiveBPFactor, SIRSFactor, SIS_heterogeneousFactor, SISFactor, glauber_factors)
from .models import SIS, Glauber, Ising
from .mpbp import (CB_BP, MPBP, random_message, periodic_mpbp, periodic_mpbp_infinite_graph, is_periodic, autocorrelations, autocovariances, belief_train, beliefs_tu, twovar_marginals, IndexedBiDiGraph, InfiniteBipartiteRegularGraph, InfiniteRegularGraph, TruncBond,
                   TruncBondMax, TruncBondThresh, TruncThresh, beliefs, bethe_free_energy, color_classes,
                   default_truncator, iterate, means, mpbp, mpbp_infinite_bipartite_graph, mpbp_infinite_graph,
                   onebpiter, pair_beliefs, reset_messages, pair_beliefs_as_mpem, pair_correlations,
                   alternate_marginals, alternate_correlations, expectation, logprob, reset, reset_observations,
                   is_free_dynamics)
