"""Model registry with the reference's names and error convention (core/networks/__init__.py:6-19)."""
from .keypoints_generation.generator import SequenceGeneratorCNN
from .keypoints_generation.discriminator import PoseSequenceDiscriminator
from .poses_reconstruction.autoencoder import Autoencoder, PoseSeqEncoder

module_dict = {
    'SequenceGeneratorCNN': SequenceGeneratorCNN,
    'PoseSequenceDiscriminator': PoseSequenceDiscriminator,
    'Autoencoder': Autoencoder,
    'PoseSeqEncoder': PoseSeqEncoder,
}


def get_model(name: str):
    try:
        return module_dict[name]
    except KeyError:
        raise KeyError('Unknown model: %s' % name) from None
