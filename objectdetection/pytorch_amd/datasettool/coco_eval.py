"""COCO-style result writer over the device decode path.

Mirrors `datasettool/coco_eval.py:6-80` of the reference: every image of `dataset` goes through
`model(img, return_loss=False)` (B = 1, the fused decode + NMS launch sequence), the boxes are divided by the
sample's resize scale, turned into (x, y, w, h) and written as `{set_name}_bbox_results.json` records.
One transfer per image brings the kept rows to the host (the reference moves three tensors and indexes
them per detection); the scale division is the host's, bit-identical to the reference's.  Scoring itself is pycocotools'
job (`COCOeval`, not installed in this image): `evaluate_coco` calls it when it is importable and returns
the records otherwise.
"""
import json

import torch


def image_results(image_id, scores, labels, boxes, scale, label_to_coco_label, threshold=0.05):
    """coco_eval.py:29-56 for one image.  scores [K] sorted descending, labels [K] int64, boxes [K,4] xyxy (device
    or host tensors).  Rows from the first score below `threshold` on are dropped (`break`, :44-45)."""
    if boxes.shape[0] == 0:
        return []
    rows = torch.cat([boxes.to(torch.float32), scores.to(torch.float32).reshape(-1, 1),
                      labels.to(torch.float32).reshape(-1, 1)], dim=1).cpu()          # one transfer per image
    b = rows[:, :4] / scale                                   # :30, on the host: a true division like the reference's
    b[:, 2] -= b[:, 0]                                        # :34-35
    b[:, 3] -= b[:, 1]
    out = []
    for (x, y, w, h), (s, lab) in zip(b.tolist(), rows[:, 4:].tolist()):
        if s < threshold:
            break
        out.append({'image_id': image_id, 'category_id': label_to_coco_label(int(lab)), 'score': float(s),
                    'bbox': [x, y, w, h]})
    return out


def collect_results(dataset, model, threshold=0.05, device=None):
    """coco_eval.py:8-60: (results, image_ids) for the whole dataset; the model is left in eval mode."""
    model.eval()
    device = next(model.parameters()).device if device is None else device
    results, image_ids = [], []
    with torch.no_grad():
        for index in range(len(dataset)):
            data = dataset[index]
            img = data['img']
            if not torch.is_tensor(img):
                img = torch.from_numpy(img)
            img = img.permute(2, 0, 1).to(device).float().unsqueeze(dim=0)       # :21
            scores, labels, boxes = model(img, return_loss=False)
            results += image_results(dataset.image_ids[index], scores, labels, boxes, data['scale'],
                                     dataset.label_to_coco_label, threshold)
            image_ids.append(dataset.image_ids[index])
    return results, image_ids


def evaluate_coco(dataset, model, threshold=0.05):
    """Drop-in for coco_eval.py:6: writes `{set_name}_bbox_results.json`; runs COCOeval when pycocotools exists.
    Returns the records (the reference returns None)."""
    results, image_ids = collect_results(dataset, model, threshold)
    if not len(results):
        model.train()
        return results
    path = '{}_bbox_results.json'.format(dataset.set_name)
    with open(path, 'w') as f:
        json.dump(results, f, indent=4)
    try:
        from pycocotools.cocoeval import COCOeval
    except ImportError:
        model.train()
        return results
    coco_true = dataset.coco
    coco_pred = coco_true.loadRes(path)
    coco_eval = COCOeval(coco_true, coco_pred, 'bbox')
    coco_eval.params.imgIds = image_ids
    coco_eval.evaluate()
    coco_eval.accumulate()
    coco_eval.summarize()
    model.train()
    return results
