"""sim.py of the reference, all games of a call run concurrently on the GPU."""
from sigma_zero_amd.sim import play_game, generate_training_data, play_games, device  # noqa: F401
