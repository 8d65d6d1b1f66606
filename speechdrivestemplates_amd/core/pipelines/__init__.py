"""Pipeline registry (core/pipelines/__init__.py:5-16)."""


def get_pipeline(name: str):
    from .voice2pose import Voice2Pose
    from .pose2pose import Pose2Pose
    table = {'Voice2Pose': Voice2Pose, 'Pose2Pose': Pose2Pose}
    try:
        return table[name]
    except KeyError:
        raise KeyError('Unknown pipeline: %s' % name) from None
