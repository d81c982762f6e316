"""mctsnode.py of the reference (read-only view class; the tree lives in HBM)."""
from sigma_zero_amd.mctsnode import Node  # noqa: F401
