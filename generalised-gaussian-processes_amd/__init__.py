"""MI355X-native sparse-GP inference core (collapsed / VFE bound) behind the reference's model-class API.

The directory name follows the project naming (``generalised-gaussian-processes_amd``), which is not a
valid Python identifier: import it through the repo-root shim ``ggp_amd`` (``import ggp_amd``).
"""
from ._lib import KERNEL_IDS, SgpLibraryError, SgpStatusError, load_library  # noqa: F401
from .composite import (CO2_LOG_PRIOR_SD, CompositeBayesianSparseGPR_HMC, CompositeHmcTarget, CompositeKernel, Factor,  # noqa: F401
                        co2_kernel)
from .core import CollapsedBound, HmcTarget, NotPositiveDefiniteError, SgpTimeoutError, shard_rows  # noqa: F401
from . import datasets, experiment_tools  # noqa: F401
from .gp_shim import (BernoulliLikelihood, ExactMarginalLogLikelihood, GaussianLikelihood, InducingPointKernel, MaternKernel,  # noqa: F401
                      MultivariateNormal, RBFKernel, ScaleKernel, ZeroMean, settings)
from .hmc import NUTS, SplitMix, Trace, sample_nuts, sample_nuts_device  # noqa: F401
from .metrics import nlpd, nlpd_marginal, nlpd_mixture, rmse  # noqa: F401
from .models import (BayesianSparseGPR_HMC, BayesianStochasticVariationalGP, SparseGPR, StochasticVariationalGP,  # noqa: F401
                     VariationalHyperDist, mixture_posterior_predictive)


def __getattr__(name):  # lazy: importing the package must work without a GPU (build / symbol checks)
    if name == "HipEngine":
        from .engine import HipEngine
        return HipEngine
    raise AttributeError(name)
