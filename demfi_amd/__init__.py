"""demfi_amd -- MI355X-native DeMFI-Net_rb inference forward (HIP kernels behind the reference's nn.Module surface)."""
from .spec import HyperParams                      # noqa: F401
from .model import DeMFInet                        # noqa: F401
from .weights import synthetic_state_dict, synthetic_window   # noqa: F401
