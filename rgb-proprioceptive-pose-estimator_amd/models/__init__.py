from .losses import PoseDistanceLoss  # noqa: F401
from .naive import NaiveEndEffectorStateEstimator, NaiveObjectStateEstimator  # noqa: F401
from .time_sensitive import (  # noqa: F401
    TemporallyDependentObjectStateEstimator,
    TemporallyDependentObjectStateEstimatorV2,
    TemporallyDependentStateEstimator,
)
