from var_amd.models.basic_vae import *  # noqa: F401,F403
from var_amd.models import basic_vae as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
