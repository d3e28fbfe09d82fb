# -*- coding: utf-8 -*-
"""
Stub of the `DatasetConfig` the model constructor reads
(/root/reference/emsanet/model.py:39-43; class imported from the un-vendored
`nicr_scene_analysis_datasets`, /root/reference/emsanet/data.py).  Only the attributes the hot
path touches exist: `semantic_label_list_without_void` (len(), `.classes_is_thing`,
`.classes_use_orientations`) and `scene_label_list_without_void` (len()).
"""


class LabelList(list):
    def __init__(self, names, is_thing=None, use_orientations=None):
        super().__init__(names)
        self.classes_is_thing = tuple(is_thing if is_thing is not None else [True] * len(names))
        self.classes_use_orientations = tuple(
            use_orientations if use_orientations is not None else [True] * len(names))


class DatasetConfig:
    def __init__(self, n_semantic_classes=40, n_scene_classes=10):
        self.semantic_label_list_without_void = LabelList(
            [f'class_{i}' for i in range(n_semantic_classes)])
        self.scene_label_list_without_void = LabelList(
            [f'scene_{i}' for i in range(n_scene_classes)])


def nyuv2_config() -> DatasetConfig:
    """NYUv2: 40 semantic classes (/root/reference/emsanet/weights.py:95-97); scene-class count
    follows `get_decoders`' default (/root/reference/emsanet/decoder.py:45)."""
    return DatasetConfig(40, 10)
