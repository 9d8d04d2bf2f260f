"""convnets_amd -- MI355X-native conv / batch-norm / max-pool / ReLU training path behind the layer
factory of milan-pavlovic-ai/ConvNets.  The compute lives in libcnk.so
(include/cnk.h, convnets_amd/csrc/); this package is the host-side mirror of the reference's
factory, models and training step.  There is no CPU fallback."""
import os as _os

# compute / weight-gradient / all-reduce / RCCL streams each need their own hardware queue (the HIP runtime
# maps streams onto GPU_MAX_HW_QUEUES queues, default 4, and streams sharing a queue serialise); effective
# only if this import happens before the first HIP call of the process
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

from .settings import Settings, DeviceMngr                      # noqa: F401
from .basemodel import MultiClassBaseModel                      # noqa: F401
from .resnet import ResNet, ResBlock, ResBottleneck             # noqa: F401
from .vggnet import VGGNet                                      # noqa: F401
from .convnet import ConvNet, LeNet5, AlexNet                   # noqa: F401
from .mobilenet import MobileNetV1                              # noqa: F401
from .resnext import ResNeXt                                    # noqa: F401
from .inception import InceptionNetV1                           # noqa: F401
from .squeezenet import SqueezeNet                              # noqa: F401
from .densenet import DenseNet                                  # noqa: F401
from .senet import SENet, SEResNet                              # noqa: F401
from .shufflenet import ShuffleNetV1                            # noqa: F401
from .sknet import SKNet, SKResNet                              # noqa: F401
from .graph import GraphedTrainStep                            # noqa: F401
from .optim import FlatAdam                                     # noqa: F401

MODELS = {'resnet': ResNet, 'vgg': VGGNet, 'convnet': ConvNet, 'lenet5': LeNet5, 'alexnet': AlexNet,
          'mobilenet': MobileNetV1, 'resnext': ResNeXt,
          'inception': InceptionNetV1, 'squeezenet': SqueezeNet,
          'densenet': DenseNet, 'seresnet': SEResNet, 'senet': SENet,
          'shufflenet': ShuffleNetV1, 'sknet': SKNet, 'skresnet': SKResNet}
