from var_amd.models.quant import *  # noqa: F401,F403
from var_amd.models import quant as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
