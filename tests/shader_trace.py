"""A second restatement of `trace_ray` (shaders/glsl/raytrace.comp:78-183) — scalar numpy float32, written from the GLSL text,
to be held bit for bit against the oracle's (tests/test_oracle_kat.py, K12).  Test infrastructure.

What it takes from this repository is only what GLSL leaves open and include/rt_math.h pins (the arithmetic CONTRACT, not the
algorithm): mod(x, y) = x - y * floor(x / y) with every operation rounded; length = sqrt(fma(z, z, fma(y, y, x * x)));
normalize(v) = v * (1 / length(v)); `position += direction * t` fused; and the samplers' meaning (render_data.rs:82-108):
minefield texel = floor(coordinate), a coordinate of exactly 256.0 reads the border (0); the material texel is
floor(fract((p + 128) / 256) * 256).  Everything else — the order of the comparisons, which fetch comes before which test,
the step size, the face offsets — is read off the shader here a second time.

Python loops: meant for a few hundred rays."""
from fractions import Fraction

import numpy as np

f32 = np.float32
R = 256
LIMIT = 2048
NORMAL_x, NORMAL_y, NORMAL_z = 0, 2, 4


def _bits(x):
    return int(np.array(x, dtype=np.float32).view(np.uint32))


def fma32(a, b, c):
    """round_to_nearest_even_f32(a * b + c), exactly (no double rounding): the exact value as a Fraction, then the nearest of the
    float32 candidates round it."""
    a, b, c = float(a), float(b), float(c)
    s = a * b + c
    if not np.isfinite(s):
        return f32(s)
    exact = Fraction(a) * Fraction(b) + Fraction(c)
    best = f32(float(exact))
    for cand in (np.nextafter(best, f32(-np.inf)), np.nextafter(best, f32(np.inf))):
        dc, db = abs(Fraction(float(cand)) - exact), abs(Fraction(float(best)) - exact)
        if dc < db or (dc == db and _bits(cand) % 2 == 0 and _bits(best) % 2 == 1):
            best = cand
    return f32(best)


def mod32(x, y):
    return f32(x - f32(y * np.floor(f32(x / y))))


def length3(v):
    return np.sqrt(fma32(v[2], v[2], fma32(v[1], v[1], f32(v[0] * v[0]))))


def normalize3(v):
    r = f32(f32(1.0) / length3(v))
    return [f32(v[0] * r), f32(v[1] * r), f32(v[2] * r)]


def get_step(minefield, tex_pos):
    """:78-80 — texture(minefield, tex_pos).r with unnormalised coordinates: the texel that contains the coordinate."""
    t = []
    for c in tex_pos:
        fc = np.floor(c)
        if not (fc >= 0 and fc < R):          # 256.0 (Q7) or NaN: outside the image, the border value
            return 0
        t.append(int(fc))
    return int(minefield[t[2], t[1], t[0]])


def trace_ray(materials, minefield, origin, direction, lr=(0, 0, 0)):
    """materials, minefield: [R, R, R] arrays indexed [z][y][x].  Returns a dict with the HitResult fields the shader defines,
    the iteration count, and `limit` = the loop ran out (fields then undefined in the shader)."""
    with np.errstate(all="ignore"):
        materials = np.asarray(materials).reshape(R, R, R)
        minefield = np.asarray(minefield).reshape(R, R, R)
        origin = [f32(x) for x in origin]
        direction = normalize3([f32(x) for x in direction])                                   # :83
        position = list(origin)                                                                # :85
        length_per_axis = [f32(f32(1.0) / np.abs(d)) for d in direction]                       # :88
        normals = [NORMAL_x + 1 if direction[0] > 0 else NORMAL_x,                             # :89-93
                   NORMAL_y + 1 if direction[1] > 0 else NORMAL_y,
                   NORMAL_z + 1 if direction[2] > 0 else NORMAL_z]
        muls = [f32(-1.0) if d > 0 else f32(1.0) for d in direction]                           # :94-98
        current_rotation = [f32(x) for x in lr]                                                # :104
        pos_offset = f32(R // 2)                                                               # :105
        width = f32(R)

        def tex_pos(p):
            return [mod32(f32(c + pos_offset), width) for c in p]

        current_step = get_step(minefield, tex_pos(position))                                  # :106
        step_size = (1 << current_step) // 2                                                   # :107
        normal = None
        air = None
        packed = 0
        iterations = 0
        limit = LIMIT
        while limit > 0:                                                                       # :113
            iterations += 1
            ss = f32(step_size)
            ltn = [f32(f32(f32(0.0001) + mod32(f32(f32(position[a] + pos_offset) * muls[a]), ss)) * length_per_axis[a])
                   for a in range(3)]                                                          # :119
            if ltn[0] < ltn[1]:                                                                # :120-136
                axis = 0 if ltn[0] < ltn[2] else 2
            else:
                axis = 1 if ltn[1] < ltn[2] else 2
            t = ltn[axis]
            position = [fma32(direction[a], t, position[a]) for a in range(3)]
            normal = normals[axis]
            current_step = get_step(minefield, tex_pos(position))                              # :137
            if (np.abs(f32(position[0] - current_rotation[0])) >= R // 2 or                    # :138-145
                    np.abs(f32(position[1] - current_rotation[1])) >= R // 2 or
                    np.abs(f32(position[2] - current_rotation[2])) >= R // 2):
                air = True
                break
            elif current_step <= 0:                                                            # :146-160
                air = False
                uvw = [mod32(f32(f32(c + pos_offset) / width), f32(1.0)) for c in position]
                t3 = [int(np.floor(f32(u * width))) if np.isfinite(u) else -1 for u in uvw]
                packed = int(materials[t3[2], t3[1], t3[0]]) if all(0 <= i < R for i in t3) else 0
                break
            step_size = (1 << current_step) // 2                                               # :161
            limit -= 1
        ran_out = air is None
        distance = length3([f32(origin[a] - position[a]) for a in range(3)])                   # :164
        offset_amount = f32(0.001)                                                             # :166-180
        if normal is not None:
            a = normal // 2
            position[a] = f32(position[a] + offset_amount) if normal % 2 == 0 else f32(position[a] - offset_amount)
        albedo = [f32(f32(packed >> s & 0x7F) / f32(127.0)) for s in (14, 7, 0)]               # :156-158
        return {"air": air, "normal": normal, "position": position, "distance": distance, "packed_material": packed,
                "albedo": albedo, "iterations": iterations, "limit": ran_out}
