import numpy as np


def rotate_ypr(rot, xhat, yhat, zhat):
    """Yaw/pitch/roll rotation of a basis; only the identity (rot == 0) is needed
    by the reference's cylinder beams, which is all this stand-in supports."""
    if np.any(np.asarray(rot) != 0.0):
        raise NotImplementedError("non-zero rotation not supported by the stand-in")
    return xhat, yhat, zhat
