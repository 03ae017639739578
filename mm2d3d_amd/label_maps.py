"""Class-merging tables of the reference's four datasets, by RAW LABEL ID.

The reference spells them as lists of class names and builds ``label_mapping`` (raw id -> merged class, -100 = ignore) in
each dataset's constructor (nuscenes_dataloader.py:21-57,163-171; semantic_kitti.py:17-104,190-199; a2d2.py:17-131,178-186;
virtual_kitti_dataloader.py:16-42,50-56).  Here the same merges are written once as ``{merged class: [raw ids]}``; the
merged-class order is the label order the networks are trained on (datasets/*.yaml ``seg_labels``).  Checked against the
reference's own mappings through the ``seg_label`` arrays of tests/golden/loader_*.npz.
"""
from __future__ import annotations

import numpy as np

# NuScenes-lidarseg, 17 raw classes (index = raw id)
NUSCENES_RAW = ("ignore", "barrier", "bicycle", "bus", "car", "construction_vehicle", "motorcycle", "pedestrian", "traffic_cone",
                "trailer", "truck", "driveable_surface", "other_flat", "sidewalk", "terrain", "manmade", "vegetation")
NUSCENES_MERGE = {"vehicle": [2, 3, 4, 5, 6, 9, 10], "driveable_surface": [11], "sidewalk": [13], "terrain": [14], "manmade": [15],
                  "vegetation": [16]}

# SemanticKITTI raw ids (semantic-kitti.yaml): 0 unlabeled, 1 outlier, 10 car, 11 bicycle, 13 bus, 15 motorcycle, 16 on-rails,
# 18 truck, 20 other-vehicle, 30 person, 31 bicyclist, 32 motorcyclist, 40 road, 44 parking, 48 sidewalk, 49 other-ground,
# 50 building, 51 fence, 52 other-structure, 60 lane-marking, 70 vegetation, 71 trunk, 72 terrain, 80 pole, 81 traffic-sign,
# 99 other-object, 252-259 the moving variants (car, bicyclist, person, motorcyclist, on-rails, bus, truck, other-vehicle)
SEMANTIC_KITTI_TABLE = 259 + 2  # the reference sizes the table "highest id + 2"
SEMANTIC_KITTI_MERGE = {
    "A2D2": {"car": [10, 252], "truck": [18, 258], "bike": [11, 15, 31, 32, 253, 255], "person": [30, 254], "road": [40, 60],
             "parking": [44], "sidewalk": [48], "building": [50], "nature": [70, 71, 72], "other-objects": [51, 80, 81, 99]},
    "VirtualKITTI": {"vegetation_terrain": [70, 71, 72], "building": [50], "road": [40, 60], "object": [51, 80, 81, 99],
                     "truck": [18, 258], "car": [10, 252]},
    "nuScenes": {"vehicle": [18, 258, 10, 252, 11, 15, 31, 32, 253, 255], "driveable_surface": [40, 60, 44], "sidewalk": [48],
                 "terrain": [72], "manmade": [50, 51, 80, 81, 99], "vegetation": [70, 71]},
}

# VirtualKITTI (vkitti3D), 14 raw classes; raw label 99 is folded into the last one ("Don't care") before the merge
VIRTUAL_KITTI_RAW = ("Terrain", "Tree", "Vegetation", "Building", "Road", "GuardRail", "TrafficSign", "TrafficLight", "Pole", "Misc",
                     "Truck", "Car", "Van", "Don't care")
VIRTUAL_KITTI_MERGE = {"vegetation_terrain": [0, 1, 2], "building": [3], "road": [4], "object": [6, 7, 8, 9], "truck": [10], "car": [11]}

# A2D2: the raw ids are positions in the dataset's own class_list.json, so the merge is given by NAME patterns
# ("Car " = every class whose name starts with it: "Car 1" .. "Car 4")
A2D2_MERGE = {
    "car": ["Car ", "=Ego car"],
    "truck": ["Truck "],
    "bike": ["Bicycle ", "Small vehicles "],
    "person": ["Pedestrian "],
    "road": ["=RD normal street", "=Zebra crossing", "=Solid line", "=RD restricted area", "=Slow drive area", "=Drivable cobblestone",
             "=Dashed line", "=Painted driv. instr."],
    "parking": ["=Parking area"],
    "sidewalk": ["=Sidewalk", "=Curbstone"],
    "building": ["=Buildings"],
    "nature": ["=Nature object"],
    "other-objects": ["=Poles", "Traffic signal ", "Traffic sign ", "=Sidebars", "=Speed bumper", "=Irrelevant signs", "=Road blocks",
                      "=Obstacles / trash", "=Animals", "=Signal corpus", "=Electronic traffic", "=Traffic guide obj.", "=Grid structure"],
}


def a2d2_match(name, patterns):
    return any(name == p[1:] if p.startswith("=") else name.startswith(p) for p in patterns)


def merged(categories, table_size):
    """(label_mapping int64 [table_size] with -100 for every raw id outside the merge, merged class names)."""
    mapping = -100 * np.ones(table_size, dtype=int)
    for idx, ids in enumerate(categories.values()):
        mapping[list(ids)] = idx
    return mapping, list(categories.keys())
