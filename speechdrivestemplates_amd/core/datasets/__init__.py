from .gesture_dataset import GestureDataset, SyntheticGestureDataset

module_dict = {'GestureDataset': GestureDataset, 'SyntheticGestureDataset': SyntheticGestureDataset}


def get_dataset(name: str):
    try:
        return module_dict[name]
    except KeyError:
        raise KeyError('Unknown dataset: %s' % name) from None
