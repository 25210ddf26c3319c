"""Writes raytracing_weekend_amd/host/assets/earthmap.ppm: a synthetic equirectangular "planet" map (P6, 512x256).

Scenes 2 and 4 of the reference texture a sphere with assets/earthmap.jpg, decoded by its vendored stb_image. This
build's host reads PPM only and ships this procedurally generated stand-in of the same shape (longitude x latitude,
blue oceans, green/brown continents, white caps). To render with the real map, convert the reference's JPEG to PPM
with any tool and point RTW_ASSET_DIR at its directory."""
import os
import numpy as np

W, H = 512, 256
rs = np.random.RandomState(20240607)
lon = (np.arange(W) + 0.5) / W * 2 * np.pi
lat = ((np.arange(H) + 0.5) / H - 0.5) * np.pi
lo, la = np.meshgrid(lon, lat)
x, y, z = np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)
height = np.zeros((H, W))
for octave in range(1, 7):  # band-limited noise on the sphere: a sum of random plane waves, seamless in longitude
    for _ in range(6):
        k = rs.normal(size=3)
        k *= (2.0 ** octave) / np.linalg.norm(k)
        height += np.sin(k[0] * x + k[1] * y + k[2] * z + rs.uniform(0, 2 * np.pi)) / (1.7 ** octave)
height /= np.abs(height).max()
land = height > 0.08
img = np.zeros((H, W, 3))
img[...] = np.array([0.05, 0.15, 0.45]) + np.clip(height[..., None] + 0.5, 0, 1) * np.array([0.02, 0.10, 0.25])
green, brown = np.array([0.15, 0.45, 0.12]), np.array([0.45, 0.35, 0.20])
t = np.clip((height - 0.08) / 0.5, 0, 1)[..., None]
img[land] = (green * (1 - t) + brown * t)[land]
cap = np.abs(la) > (1.25 - 0.15 * height)
img[cap] = 0.93
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing_weekend_amd", "host", "assets", "earthmap.ppm")
with open(out, "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % (W, H))
    f.write((np.clip(img[::-1], 0, 1) * 255 + 0.5).astype(np.uint8).tobytes())  # file rows run north to south
print(out, os.path.getsize(out))
