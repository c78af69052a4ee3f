"""phoenix_amd -- MI355X-native NeuralODE integration engine for PHOENIX's ODENet.

Drop-in for the reference's hot path only:  `from phoenix_amd import odeint_adjoint as odeint`
replaces `from torchdiffeq import odeint_adjoint as odeint` (train_insilico.py:15-18)."""
from .odeint import SOLVERS, odeint, odeint_adjoint, odeint_calls, odeint_per_sample  # noqa: F401
from .odenet import ODENet  # noqa: F401
from .engine import check_pending_status, set_status_mode  # noqa: F401
from .training import training_step  # noqa: F401
from .graphs import GraphedStep  # noqa: F401
from .data import DataHandler, readcsv, writecsv  # noqa: F401
from .prior import PriorMatrix, prior_targets, read_prior_matrix  # noqa: F401
from .analysis import gene_influence_scores  # noqa: F401
from .simulator import HillSystem, generate_dataset  # noqa: F401

__version__ = "0.1.0"
