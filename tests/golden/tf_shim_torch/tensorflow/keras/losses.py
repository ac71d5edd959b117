import numpy as _np


def _mean_last(v):
    import tensorflow as tf
    return tf.reduce_mean(v, axis=-1)


def MSE(y_true, y_pred):  # noqa: N802
    import tensorflow as tf
    return _mean_last(tf.square(tf.convert_to_tensor(y_pred) - tf.convert_to_tensor(y_true)))


def MAE(y_true, y_pred):  # noqa: N802
    import tensorflow as tf
    return _mean_last(tf.abs(tf.convert_to_tensor(y_pred) - tf.convert_to_tensor(y_true)))


class _Loss:
    fn = None

    def __init__(self, reduction='none'):
        assert reduction == 'none', 'only reduction="none" is used by the reference'

    def __call__(self, y_true, y_pred, sample_weight=None):
        v = type(self).fn(y_true, y_pred)
        if sample_weight is not None:
            v = v * sample_weight
        return v


class MeanSquaredError(_Loss):
    fn = staticmethod(MSE)


class MeanAbsoluteError(_Loss):
    fn = staticmethod(MAE)
