"""Import-path aliases of the drop-in classes under the reference's module names (``models.networks`` ...): the reference's
scripts import ``from models.networks import VaeGan`` etc.  The implementations live in ``vae_play_amd``.  (Checkpoints are exchanged
as ``state_dict``s -- vae_play_amd/checkpoint.py -- never as pickled module objects.)"""
