"""`tensorflow` as the reference's DRIVER SCRIPTS see it once the hot path runs on the MI355X library: train.py:6,51,273, run_eval.py:12,122 and vae/train_vae.py:10
import it and touch exactly two symbols -- tf.random.set_random_seed(seed) and tf.reset_default_graph() -- everything else that used TensorFlow lived in
vae/models.py, ppo.py and utils.py, which this directory replaces.  With carla-ppo_amd/ ahead of the reference checkout on PYTHONPATH the unchanged scripts import
this module and INTEGRATION.md's command line works without TensorFlow installed.

  tf.random.set_random_seed(seed)   the graph-level seed of the reference (train.py:50-51): here the default seed of the engines' own noise sources
                                    (mi_vae_set_seed / the PPO exploration noise): mi355.seed.graph_seed() returns it to models created afterwards
  tf.reset_default_graph()          nothing to reset: there is no global graph (train.py:273)
Anything else raises AttributeError naming this file, so that a script that really needs TensorFlow fails loudly instead of half-working.
"""
import types

_graph_seed = [None]


def _set_random_seed(seed):
    _graph_seed[0] = None if seed is None else int(seed)


def get_graph_seed():
    """The last tf.random.set_random_seed value (None: unseeded, as TensorFlow's default)."""
    return _graph_seed[0]


def reset_default_graph():
    return None


random = types.SimpleNamespace(set_random_seed=_set_random_seed)
set_random_seed = _set_random_seed                    # TF 1.x also exposed it at the top level
__version__ = "0.0-mi355-stub"


def __getattr__(name):
    raise AttributeError("tensorflow stub of carla-ppo_amd (%s): the driver scripts only need tf.random.set_random_seed and tf.reset_default_graph; "
                         "`tf.%s` is not provided -- the TensorFlow graph code lives in vae/models.py / ppo.py / utils.py, which the MI355X modules replace" % (__file__, name))
