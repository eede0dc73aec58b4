from var_amd.models.basic_var import *  # noqa: F401,F403
from var_amd.models import basic_var as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
