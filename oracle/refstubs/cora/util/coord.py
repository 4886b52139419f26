import numpy as np


def sph_to_cart(sph_arr):
    sph_arr = np.asarray(sph_arr)
    theta, phi = sph_arr[..., 0], sph_arr[..., 1]
    out = np.empty(sph_arr.shape[:-1] + (3,), dtype=np.float64)
    st = np.sin(theta)
    out[..., 0] = st * np.cos(phi)
    out[..., 1] = st * np.sin(phi)
    out[..., 2] = np.cos(theta)
    return out


def sph_dot(a, b):
    return np.sum(sph_to_cart(a) * sph_to_cart(b), axis=-1)


def thetaphi_plane_cart(sph_arr):
    sph_arr = np.asarray(sph_arr)
    theta, phi = sph_arr[..., 0], sph_arr[..., 1]
    that = np.empty(sph_arr.shape[:-1] + (3,), dtype=np.float64)
    phat = np.empty(sph_arr.shape[:-1] + (3,), dtype=np.float64)
    that[..., 0] = np.cos(theta) * np.cos(phi)
    that[..., 1] = np.cos(theta) * np.sin(phi)
    that[..., 2] = -np.sin(theta)
    phat[..., 0] = -np.sin(phi)
    phat[..., 1] = np.cos(phi)
    phat[..., 2] = 0.0
    return that, phat


def norm_vec2(vec2):
    n = np.hypot(vec2[..., 0], vec2[..., 1])
    n = np.where(n == 0.0, 1.0, n)
    vec2[..., 0] /= n
    vec2[..., 1] /= n
