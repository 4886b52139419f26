"""Natural cubic spline (second derivative zero at both ends), the published
algorithm behind cora.util.cubicspline.Interpolater."""
import numpy as np


class Interpolater(object):
    def __init__(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        n = x.size
        y2 = np.zeros(n)
        u = np.zeros(n)
        for i in range(1, n - 1):
            sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1])
            p = sig * y2[i - 1] + 2.0
            y2[i] = (sig - 1.0) / p
            u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (
                x[i] - x[i - 1]
            )
            u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p
        for k in range(n - 2, -1, -1):
            y2[k] = y2[k] * y2[k + 1] + u[k]
        self.x, self.y, self.y2 = x, y, y2

    def __call__(self, xv):
        xv = np.asarray(xv, dtype=np.float64)
        x, y, y2 = self.x, self.y, self.y2
        khi = np.clip(np.searchsorted(x, xv, side="left"), 1, x.size - 1)
        klo = khi - 1
        h = x[khi] - x[klo]
        a = (x[khi] - xv) / h
        b = (xv - x[klo]) / h
        return (
            a * y[klo]
            + b * y[khi]
            + ((a**3 - a) * y2[klo] + (b**3 - b) * y2[khi]) * (h * h) / 6.0
        )
