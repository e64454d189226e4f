from .convert import to_gpu, to_cpu, set_device
from .functions import DiffLoss, CMD, ReverseLayerF, getBinaryTensor
from .eval import get_accuracy, get_metrics, DeviceEval
