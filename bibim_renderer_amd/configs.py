"""Workload definitions: BASELINE.json configs C1..C5 as plain parameters (no arithmetic here).

Numbers follow SURVEY.md section 8(d).  Instance i of an n = G*G grid sits at
translate(2*(i%G) - (G-1), -1, 2 + 2*(i//G)) * rotateY(-90) * rotateX(-90) * scale(0.01)
(reference per-instance chain: src/scene.cpp:180-187); the reference's own single-instance layout
(translate(2i, -1, 2)) is kept for n = 1.
"""
from __future__ import annotations

from dataclasses import dataclass, field

# reference light colours: src/scene.cpp:27 and :33
_COLOR_A = (1.0, 0.8, 0.8)
_COLOR_B = (0.8, 1.0, 0.8)


@dataclass(frozen=True)
class PointLight:
    pos: tuple
    color: tuple
    intensity: float = 50.0


@dataclass(frozen=True)
class Config:
    name: str
    width: int
    height: int
    grid: int                      # G: G*G ShaderBall instances (1 => reference layout)
    cam_pos: tuple
    cam_yaw: float
    cam_pitch: float
    lights: tuple = field(default_factory=tuple)
    enable_normal_map: int = 1
    fov: float = 60.0              # src/main.cpp:1331-1332
    near: float = 0.1
    far: float = 1000.0
    texture_size: int = 2048
    description: str = ""

    @property
    def n_instances(self):
        return self.grid * self.grid

    def scaled(self, width, height, texture_size=None):
        """Same scene at another resolution (parity tests run reduced sizes)."""
        from dataclasses import replace
        return replace(self, width=width, height=height,
                       texture_size=texture_size or self.texture_size,
                       name=f"{self.name}@{width}x{height}")


def _grid_lights(n, g):
    # (2*(j%G') - (G'-1), 2, 2 + 4*(j//G')), alternating the two reference colours
    out = []
    for j in range(n):
        out.append(PointLight((2.0 * (j % g) - (g - 1), 2.0, 2.0 + 4.0 * (j // g)), _COLOR_A if j % 2 == 0 else _COLOR_B))
    return tuple(out)


C2 = Config("C2", 1920, 1080, 1, (0.0, 0.0, 0.0), 0.0, 0.0, (PointLight((0.0, 2.0, 0.0), _COLOR_A),),
            description="ShaderBall x1 + plane, 1 point light, GGX PBR + normal map, 1920x1080")
C3 = Config("C3", 3840, 2160, 4, (0.0, 2.0, -2.0), 0.0, -15.0, _grid_lights(4, 2),
            description="ShaderBall x16 + plane, 4 point lights, GGX PBR + normal map, 3840x2160")
C4 = C3  # same frame, screen-band split over 2/4/8 GPUs
C5 = Config("C5", 7680, 4320, 8, (0.0, 4.0, -6.0), 0.0, -20.0, _grid_lights(8, 4),
            description="ShaderBall x64 + plane, 8 point lights, GGX PBR + normal map, 7680x4320")

CONFIGS = {"c2": C2, "c3": C3, "c4": C4, "c5": C5}
