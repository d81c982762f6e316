"""torchrun entry for sigma_zero_amd.train_rl.main (the package directory name has a hyphen, so `-m` cannot reach it)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_zero_amd.train_rl import main

if __name__ == "__main__":
    main()
