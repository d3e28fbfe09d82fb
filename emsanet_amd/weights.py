# -*- coding: utf-8 -*-
"""
Checkpoint loading with the reference's state-dict surgery.

`load_weights(args, model, state_dict)` stands in for /root/reference/emsanet/weights.py:11-162:
the same cases, decided on the same key fragments, ending in `load_state_dict(strict=True)`
(:162).  Cases (reference lines in brackets):
  1. renamed prefix `fused_encoders.*` -> `encoder.*`                                   [22-26]
  2. checkpoint trained WITH orientation, model without: drop `task_convs.2`, cut the shared
     conv / its norm from 96 to 64 channels, keep the first 3 of 5 shared depth-wise upsampling
     channels                                                                            [28-56]
  3. semantic-only model fed a panoptic checkpoint: `decoders.panoptic_helper.semantic_decoder.`
     -> `decoders.semantic_decoder.`                                                     [58-66]
  4. keys the model does not have are dropped                                            [68-78]
  5. scene head with a different number of classes keeps the model's own weights         [80-91]
  6. semantic head 37 (SUNRGB-D) <-> 40 (NYUv2/ScanNet/Hypersim) classes: copy / keep the
     first 37 channels; any remaining shape mismatch keeps the model's weights   [93-119,147-160]
  7. ScanNet, not in benchmark mode: a head trained on the 40 (549) dataset classes is cut down
     to the 20 (200) benchmark classes with the dataset's class mapping                [121-145]
     The mapping tables (`ScanNet.SEMANTIC_CLASSES_40_MAPPING_TO_BENCHMARK`, `..._549_..._200`)
     live in the un-vendored `nicr_scene_analysis_datasets` package, so the mapping is PASSED IN
     (`scannet_mapping={class_in_dataset: class_in_benchmark}`, void = key 0, ignored = value 0 --
     the structure the reference iterates over); without it only the checkpoints that need the
     table (40 / 549 channels into a smaller head) raise, everything else falls through to the
     keep-the-model's-weights rule like in the reference.
"""
import torch


def _has(key, *fragments):
    return all(f in key for f in fragments)


def load_weights(args, model, state_dict, verbose=True, scannet_mapping=None):
    log = print if verbose else (lambda *a, **k: None)
    own = model.state_dict()
    sd = {k.replace('fused_encoders.', 'encoder.'): v for k, v in state_dict.items()}   # case 1

    tasks = tuple(args.tasks)
    if 'instance' in tasks and 'orientation' not in tasks:                               # case 2
        if any(_has(k, 'instance_decoder', 'head', 'task_convs.2') for k in sd):
            log("Detected pretrained weights with orientation, removing orientation weights "
                "in instance head.")
            for k in list(sd):
                v = sd[k]
                if _has(k, 'instance_decoder', 'head', 'shared_conv'):
                    if v.dim() > 0 and v.shape[0] == 96:
                        sd[k] = v[:-32]
                elif _has(k, 'instance_decoder', 'head', 'task_convs.2'):
                    del sd[k]
                elif _has(k, 'instance_decoder', 'head', 'upsampling'):
                    sd[k] = v[:3]

    if len(tasks) == 1 and tasks[0] == 'semantic':                                       # case 3
        sd = {k.replace('decoders.panoptic_helper.semantic_decoder.',
                        'decoders.semantic_decoder.'): v for k, v in sd.items()}

    if len(sd) != len(own):                                                              # case 4
        for k in list(sd):
            if k not in own:
                log(f"Removing '{k}' from loaded state dict as the current model does not "
                    "contain such key.")
                sd.pop(k)

    for k in list(sd):                                                                   # case 5
        if _has(k, 'scene_decoder', 'head') and k in own and sd[k].shape[0] != own[k].shape[0]:
            log(f"Skipping '{k}' as the number of scene classes differs "
                f"{own[k].shape[0]} (current) vs. {sd[k].shape[0]} (pretraining).")
            sd[k] = own[k]

    if 'semantic' in tasks:                                                              # case 6
        dataset = getattr(args, 'dataset', 'nyuv2')
        sem = [k for k in sd if _has(k, 'semantic_decoder', 'head', 'conv') and k in own]
        if dataset.startswith('nyuv2'):
            for k in sem:
                if sd[k].shape[0] == 37 and own[k].shape[0] == 40:
                    log(f"Reusing 37/40 channels in '{k}'.")
                    merged = own[k].clone()
                    merged[:37] = sd[k]
                    sd[k] = merged
        if dataset.startswith('sunrgbd'):
            for k in sem:
                if sd[k].shape[0] == 40 and own[k].shape[0] == 37:
                    log(f"Removing last 3 channels in '{k}'.")
                    sd[k] = sd[k][:37]
        elif (dataset.startswith('scannet')
              and not getattr(args, 'validation_scannet_benchmark_mode', False)):
            mapping = scannet_mapping if scannet_mapping is not None \
                else getattr(args, 'scannet_semantic_mapping', None)
            if mapping is not None:
                mask = torch.tensor([c_benchmark != 0                   # class is not ignored
                                     for c_data, c_benchmark in mapping.items()
                                     if c_data != 0], dtype=torch.bool)  # skip void class
                for k in sem:
                    if sd[k].shape[0] == mask.shape[0]:
                        log(f"Removing channels for ignored classes in '{k}'.")
                        sd[k] = sd[k][mask.to(sd[k].device)]
            else:
                need = [k for k in sem if sd[k].shape[0] in (40, 549)
                        and sd[k].shape[0] != own[k].shape[0]]
                if need:
                    raise NotImplementedError(
                        f"'{need[0]}' has {sd[need[0]].shape[0]} classes: cutting it down to the "
                        "ScanNet benchmark classes needs the class mapping of "
                        "nicr_scene_analysis_datasets -- pass it as scannet_mapping=")
        for k in sem:
            if sd[k].shape != own[k].shape:
                log(f"Removing '{k}' from loaded state dict as the shape does not match: "
                    f"{tuple(sd[k].shape)} vs. {tuple(own[k].shape)}.")
                sd[k] = own[k]

    model.load_state_dict(sd, strict=True)
    return model


def load_backbone_weights(backbone, filepath, modality, verbose=True):
    """`get_backbone(..., pretrained=True, pretrained_filepath=...)` of the reference
    (/root/reference/emsanet/model.py:47-74, args.py:119-123,174,207,232): the ImageNet-pretrained
    ResNet-NBt1D weights of ONE backbone from a checkpoint file.  The loader of the reference lives in
    the un-vendored nicr_mt_scene_analysis, so the file layout is [U]; accepted here: a plain state
    dict or one under 'state_dict' / 'model', keys optionally prefixed ('module.', 'backbone.',
    'encoder.backbone_<modality>.'), with the backbone's own key names.  A 3-channel
    stem in front of a 1-channel (depth) backbone is summed over its input channels, a 1- or
    3-channel stem in front of the 4-channel rgbd stem is refused.  Classifier keys (`fc.*`) are
    dropped.  Everything else must match: a file that fills less than the whole backbone raises."""
    import torch
    log = print if verbose else (lambda *a, **k: None)
    sd = torch.load(filepath, map_location='cpu')
    for wrap in ('state_dict', 'model'):
        if isinstance(sd, dict) and wrap in sd and isinstance(sd[wrap], dict):
            sd = sd[wrap]
    own = backbone.state_dict()
    clean = {}
    for k, v in sd.items():
        for pre in ('module.', f'encoder.backbone_{modality}.', 'backbone.'):
            if k.startswith(pre):
                k = k[len(pre):]
        if k.startswith('fc.') or k not in own:
            continue
        if k == 'conv1.weight' and v.shape[1] != own[k].shape[1]:
            if v.shape[1] == 3 and own[k].shape[1] == 1:
                log(f"backbone '{modality}': summing the pretrained 3-channel stem over its input channels")
                v = v.sum(1, keepdim=True)
            else:
                raise NotImplementedError(f"pretrained stem with {v.shape[1]} input channels for a "
                                          f"{own[k].shape[1]}-channel '{modality}' backbone")
        if tuple(v.shape) != tuple(own[k].shape):
            raise RuntimeError(f"{filepath}: '{k}' has shape {tuple(v.shape)}, the backbone {tuple(own[k].shape)}")
        clean[k] = v
    missing = [k for k in own if k not in clean and not k.endswith('num_batches_tracked')]
    if missing:
        raise RuntimeError(f"{filepath}: {len(missing)} of {len(own)} backbone tensors missing "
                           f"(first: {missing[0]}) -- not a ResNet-NBt1D checkpoint of this layout")
    backbone.load_state_dict(clean, strict=False)
    log(f"backbone '{modality}': loaded {len(clean)} pretrained tensors from {filepath}")
    return backbone
