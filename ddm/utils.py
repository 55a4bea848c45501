from adm_amd.ddm.utils import *  # noqa: F401,F403
from adm_amd.ddm.utils import construct_class_by_name, get_obj_by_name, default, exists, cycle  # noqa: F401
