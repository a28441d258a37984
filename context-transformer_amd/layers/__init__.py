from .functions import *
from .modules import *
