"""Other racket outlines for the parity tests (the product's comes from racket.stl: 38 edges)."""
import numpy as np


def with_outline(p, n_edges, radius_y=0.13, radius_z=0.33):
    """`p` with another racket outline: a convex n-gon (an ellipse sampled at uneven angles), CCW in (y, z) about the COM, with the
    fields params.default_params derives from the outline set the same way"""
    from tennisbot_rl_amd.params import hull_edge_table
    rng = np.random.default_rng(100 + n_edges)
    ang = np.sort((np.arange(n_edges) + rng.uniform(-0.3, 0.3, n_edges)) * (2.0 * np.pi / n_edges))
    verts = np.stack([radius_y * np.cos(ang), radius_z * np.sin(ang)], 1)
    rec = hull_edge_table(verts, (0.0, 0.0), 1.0)
    q = p.copy()
    q.n_hull = n_edges
    e = np.ctypeslib.as_array(q.hull_edges)
    e[:] = 0.0
    e[:n_edges] = rec
    vmax = float(np.sqrt((rec[:, :2].astype(np.float64) ** 2).sum(1).max() + float(q.racket_half_thick) ** 2))
    q.hull_bound_radius = vmax * 1.0001
    q.racket_ground_threshold = 0.02 * vmax
    return q
