"""brdf.renderer — light-sphere geometry (reference: brdf/renderer.py:184-219).  Host-side NumPy
(float64), evaluated once per model; the per-ray work happens in libnfx."""
import numpy as np


def gen_light_xyz(envmap_h, envmap_w, envmap_radius=1e2):
    """Positions of the pixels of a latitude-longitude environment map on a sphere of radius
    `envmap_radius`, plus the solid angle each pixel subtends (sums to 4*pi).

    Latitudes run from just below +pi/2 (top row) to just above -pi/2, longitudes from just below
    +pi (first column) to just above -pi: the polar / seam samples are excluded by shrinking the
    grid by one step of an (h+2) x (w+2) grid."""
    lat_step = np.pi / (envmap_h + 2)
    lng_step = 2 * np.pi / (envmap_w + 2)
    lat = np.linspace(np.pi / 2 - lat_step, -np.pi / 2 + lat_step, envmap_h)
    lng = np.linspace(np.pi - lng_step, -np.pi + lng_step, envmap_w)
    lng, lat = np.meshgrid(lng, lat)
    xyz = envmap_radius * np.stack(
        (np.cos(lat) * np.cos(lng), np.cos(lat) * np.sin(lng), np.sin(lat)), axis=-1)
    sin_colat = np.sin(np.pi / 2 - lat)
    areas = 4 * np.pi * sin_colat / np.sum(sin_colat)
    if np.any(areas == 0):
        raise ValueError("There shouldn't be light pixel that doesn't contribute")
    return xyz, areas
