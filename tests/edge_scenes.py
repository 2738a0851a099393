"""Scenes that exercise the corners of TraceRay (main.cpp:84-188) the default scene never reaches:
planes as emitters and the alternate (black) tile material, distance ties between spheres and between
a sphere and a plane, several planes, negative plane distances, emitters that are the hit object,
nested/touching spheres, a camera inside a sphere, huge and far objects.

Shared by tests/gen_golden.py (which renders them through the REFERENCE'S OWN TraceRay, oracle/_ref)
and by the CPU and GPU parity tests.  Data only: object lists and the sizes/depths they are pinned at.
"""

# (max_depth, (width, height)) pairs every scene is pinned at; spp 1 and 4 (the reference's patterns)
CASES = [(5, (203, 117)), (0, (64, 40)), (15, (96, 56)), (1, (9, 7))]
CAMERA = ((0, 6, 8), (0, -.8, -1), 60.0)        # InitScene's camera, main.cpp:23-26


def edge_scenes(capi):
    M = capi.material
    light = M((1, 1, 1), (0, 0, 0), 0.0, (1, 1, 1))
    red = M((.7, .1, .1), (.9, .1, .1), 0.5)
    mirror = M((.2, .2, .2), (1, 1, 1), 0.9)
    glow = M((.3, .6, .2), (.5, .5, .5), 0.4, (.2, .1, .4))     # reflective AND emissive
    S, P = capi.sphere, capi.tiled_plane
    return {
        "plane_only": [P((0, 0, 0), (0, 1, 0))],
        "one_sphere_no_light": [S(red, (0, 1, 0), 1.0)],
        "odd_sphere_count": [S(red, (0, 1, 0), 1.0), S(light, (4, 6, 3), 0.5), S(mirror, (-2.2, 1, 1), 1.0), P((0, 0, 0), (0, 1, 0))],
        "plane_first_in_list": [P((0, 0, 0), (0, 1, 0)), S(red, (0, 1, 0), 1.0), S(light, (4, 6, 3), 0.5)],
        "two_planes": [S(light, (0, 5, 0), 0.5), P((0, 0, 0), (0, 1, 0)), P((0, 0, -12), (0, 0, 1)), S(mirror, (0, 1.5, -3), 1.5)],
        "emissive_plane": [S(red, (0, 1, 0), 1.0), P((0, 0, 0), (0, 1, 0), white=M((1, 1, 1), (1, 1, 1), 0.6, (.3, .3, .1)))],
        "coincident_spheres": [S(red, (0, 1, 0), 1.0), S(mirror, (0, 1, 0), 1.0), S(light, (3, 5, 3), 0.4), P((0, 0, 0), (0, 1, 0))],
        "nested_and_touching": [S(mirror, (0, 2, 0), 2.0), S(red, (0, 2, 0), 0.5), S(red, (4, 2, 0), 2.0), S(light, (0, 9, 4), 1.0),
                                P((0, 0, 0), (0, 1, 0))],
        "camera_inside_sphere": [S(mirror, (0, 6, 8), 3.0), S(light, (0, 6.5, 7), 0.3), P((0, 0, 0), (0, 1, 0))],
        "light_behind_camera_and_huge": [S(light, (0, 50, 100), 20.0), S(red, (0, 1, 0), 1.0), S(mirror, (1e6, 0, -1e6), 5e5),
                                          P((0, 0, 0), (0, 1, 0))],
        "mirror_hall": [S(mirror, (-3, 2, 0), 2.0), S(mirror, (3, 2, 0), 2.0), S(mirror, (0, 2, -4), 2.0), S(light, (0, 8, 2), 0.7),
                        P((0, 0, 0), (0, 1, 0), white=M((1, 1, 1), (1, 1, 1), 0.9), black=M((0, 0, 0), (0, 0, 0), 0.9))],
        # both tile materials emit (different colours: the `alt` material of a plane emitter), a second
        # emissive plane above, and a sphere that emits and reflects
        "two_emissive_planes_and_glow": [P((0, 0, 0), (0, 1, 0), white=M((1, 1, 1), (1, 1, 1), 0.6, (.3, .3, .1)),
                                           black=M((0, 0, 0), (.2, .2, .2), 0.6, (.05, .1, .3))),
                                         S(glow, (0, 1.5, 0), 1.5), S(red, (3, 1, 1), 1.0),
                                         P((0, 14, 0), (0, -1, 0), white=M((.5, .5, .5), (0, 0, 0), 0.0, (.4, .4, .4)))],
        # equal spheres listed twice with different materials, emitter first and emitter second: the
        # `occluder == emitterObj` branch against an equally near other object (main.cpp:123-149)
        "tied_emitters": [S(light, (2, 3, 0), 1.0), S(red, (2, 3, 0), 1.0), S(red, (-2, 3, 0), 1.0), S(light, (-2, 3, 0), 1.0),
                          S(mirror, (0, 1, 2), 1.0), P((0, 0, 0), (0, 1, 0))],
        # a sphere resting exactly in the plane and one cut by it; a plane facing away from the camera
        "sphere_in_plane": [S(mirror, (0, 0, 0), 1.5), S(red, (3, 1, 0), 1.0), S(light, (0, 6, 3), 0.5), P((0, 0, 0), (0, 1, 0)),
                            P((0, 20, 0), (0, 1, 0))],
    }
