"""Import-path aliases of the drop-in classes under the reference's module names (``models.networks`` ...): the reference's
checkpoints reference classes by that path (test_BE.py:79-80 unpickles module objects; train.py:154-161 pickles them), and its
scripts import ``from models.networks import VaeGan`` etc.  The implementations live in ``vae_play_amd``."""
