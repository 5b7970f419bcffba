from .MMT4Caption import MMT4Caption  # noqa: F401
from .MMEncoder import MultiModalEncoder  # noqa: F401
from .CapDecoder import CapDecoder  # noqa: F401
