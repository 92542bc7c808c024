"""The SemanticKITTI raw-label -> training-class table the dataloaders apply (reference ``src/dataset/definitions.py:3-39`` ``id_map``),
kept here as data so the device-side decode (``slu_kitti_decode``) has it where the reference tree is absent; written class-major
(training class -> the raw ids folded into it).  ``tests/test_host_logic.py`` checks it against the table stored in the golden
fixture that was generated from the reference.  In drop-in mode every other name of the reference module (colour maps, class names,
the reduced map ...) is re-exported from the shadowed file."""

_RAW_IDS_OF_CLASS = {
    0: (0, 1, 9, 52, 99),               # unlabeled (+ outlier, other-structure, other-object)
    1: (10, 252),                        # car (+ moving)
    2: (11,),                            # bicycle
    3: (15,),                            # motorcycle
    4: (18, 258),                        # truck (+ moving)
    5: (13, 16, 20, 256, 257, 259),      # other-vehicle (bus, on-rails and their moving variants)
    6: (30, 254),                        # person (+ moving)
    7: (31, 253),                        # bicyclist (+ moving)
    8: (32, 255),                        # motorcyclist (+ moving)
    9: (40,), 10: (44,), 11: (48,), 12: (49,),      # road, parking, sidewalk, other-ground
    13: (50,), 14: (51,),                # building, fence
    15: (70,), 16: (71,), 17: (72,),     # vegetation, trunk, terrain
    18: (80,),                           # pole
    19: (60, 81),                        # traffic-sign (+ lane-marking)
}
id_map = {raw: cls for cls, raws in _RAW_IDS_OF_CLASS.items() for raw in raws}

# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
