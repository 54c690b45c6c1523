"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the ray-march path (see vr_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  PARITY UNPINNED: the reference has no tests / golden vectors / CPU path.
"""
from .binding import (  # noqa: F401
    Camera,
    OracleParams,
    build_oracle,
    default_camera_block,
    gen_noise_ball,
    gen_sphere_u8,
    render,
    spline_tf,
)
