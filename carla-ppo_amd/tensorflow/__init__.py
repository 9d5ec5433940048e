"""`tensorflow` as the reference's DRIVER SCRIPTS see it once the hot path runs on the MI355X library: train.py:6,51,273, run_eval.py:12,122 and vae/train_vae.py:10
import it and touch exactly two symbols -- tf.random.set_random_seed(seed) and tf.reset_default_graph() -- everything else that used TensorFlow lived in
vae/models.py, ppo.py and utils.py, which this directory replaces.  With carla-ppo_amd/ ahead of the reference checkout on PYTHONPATH the unchanged scripts import
this module and INTEGRATION.md's command line works without TensorFlow installed.

  tf.random.set_random_seed(seed)   the graph-level seed of the reference (train.py:50-51): remembered here (get_graph_seed) and CONSUMED by mi355.init.seed_from_numpy_state --
                                    the default seed of models built without an explicit one (weight initialisation, mi_vae_set_seed, the PPO exploration noise): a script
                                    that seeds only TensorFlow gets reproducible models, as it did with the reference
  tf.reset_default_graph()          nothing to reset: there is no global graph (train.py:273)
Anything else raises AttributeError naming this file, so that a script that really needs TensorFlow fails loudly instead of half-working.
"""
import os
import sys
import types
import warnings

_graph_seed = [None]


def _warn_if_shadowing():
    """A real TensorFlow elsewhere on sys.path is hidden from EVERY importer of this process (tensorboard.compat, ...) while this directory is ahead of it: say so once, by name."""
    try:
        from importlib.machinery import PathFinder
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        other = PathFinder.find_spec("tensorflow", [p for p in sys.path if p and os.path.abspath(p) != here])
        if other is not None and other.origin and os.path.abspath(other.origin) != os.path.abspath(__file__):
            warnings.warn("carla-ppo_amd's two-symbol `tensorflow` stub (%s) shadows a real TensorFlow at %s for every import in this process; put carla-ppo_amd behind it on "
                          "PYTHONPATH (and import vae.models / ppo from carla-ppo_amd explicitly) if anything else here needs the real one" % (__file__, other.origin), RuntimeWarning)
    except Exception:
        pass


_warn_if_shadowing()


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
