"""network.py of the reference (same state_dict keys and init order)."""
from sigma_zero_amd.network import policyNN, ResidualBlock as BasicBlock  # noqa: F401
