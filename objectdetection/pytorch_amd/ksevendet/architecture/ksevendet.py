"""Drop-in for ksevendet/architecture/ksevendet.py:20-301 (`KSevenDet`): the detector `train_ksevendet.py:207`
builds from a YAML model config (config/model/ksevendet_resnet.yaml).

Same constructor (`KSevenDet(cfg, num_classes=80, iou_threshold=0.5, **kwargs)`), attribute tree and state_dict
keys (backbone.*, neck.*, regressor.*, classifier.*), same `forward(img_batch, annotations=None,
return_head=False, return_loss=False)` contract (:197-286).  Built: `backbone: resnet` (variant resnet18/34/50/
101/152), `neck: fpn` and `neck: panet-fpn` (PANetFPN, :45-59), `neck: bifpn` (BiFPN, :64-86; config/model/
ksevendet_resnet_bifpn.yaml), the simple heads with per-level BatchNorm, shared over the levels or one per level.
Other backbones and the `tfdbifpn` neck raise NotImplementedError.  The modules hold parameters; the forward is one launch
plan (engine_kseven.KSevenPlan), exposed to autograd as a single node like architectures/retinanet.py."""
import torch.nn as nn

from ... import engine_kseven, engine_kseven_bifpn
from ...architectures import retinanet as _retina
from .anchors import Anchors
from .backbone import registry
from .head import Classifier, Regressor
from .losses import FocalLoss
from .neck import BiFPN, PANetFPN
from .utils import BBoxTransform, ClipBoxes


class KSevenDet(nn.Module):
    _plan_class = engine_kseven.KSevenPlan

    def __init__(self, cfg, num_classes=80, iou_threshold=0.5, **kwargs):
        super(KSevenDet, self).__init__()
        kwargs.pop('pretrained', None)                      # train_ksevendet.py:207 passes pretrained=False
        self.convert_onnx = False
        self.fixed_size = None
        logger = kwargs.get('logger', None)
        self.num_classes = num_classes
        self.backbone_feature_pyramid_levels = cfg['backbone_feature_pyramid_levels']
        self.neck_feature_pyramid_levels = cfg['neck_feature_pyramid_levels']
        self.head_feature_pyramid_levels = cfg.get('head_feature_pyramid_levels', cfg['neck_feature_pyramid_levels'])
        self.build_config = cfg
        self.iou_threshold = iou_threshold
        self.backbone = self._build_backbone(cfg, logger=logger)
        feature_channels = self.backbone.feature_channels()
        fpn_features_num = cfg['neck_config']['features_num']
        if cfg['neck'] in ('fpn', 'panet-fpn'):             # ksevendet.py:45-59
            self.neck = PANetFPN(feature_channels, in_pyramid_levels=self.backbone_feature_pyramid_levels,
                                 out_pyramid_levels=self.neck_feature_pyramid_levels, features_num=fpn_features_num,
                                 panet_buttomup=cfg['neck'] == 'panet-fpn', logger=logger)
        elif cfg['neck'] == 'bifpn':                        # ksevendet.py:64-86
            self.neck = BiFPN(feature_channels, in_pyramid_levels=self.backbone_feature_pyramid_levels,
                              neck_pyramid_levels=self.neck_feature_pyramid_levels,
                              out_pyramid_levels=self.head_feature_pyramid_levels, features_num=fpn_features_num,
                              repeats=cfg['neck_config']['bifpn_repeats'], attention=cfg['neck_config']['bifpn_attention'],
                              logger=logger)
            self._plan_class = engine_kseven_bifpn.KSevenBiFPNPlan
        elif cfg['neck'] == 'tfdbifpn':
            raise NotImplementedError("neck 'tfdbifpn' (ksevendet/architecture/neck/bifpn.py:400-, a stack of TFDBiFPN cells) "
                                      "is not built on the MI355X path")
        else:
            raise ValueError('Unknown neck %s' % cfg['neck'])
        if list(self.backbone_feature_pyramid_levels) != [3, 4, 5] or list(self.neck_feature_pyramid_levels) != [3, 4, 5, 6, 7]:
            raise NotImplementedError('pyramid levels other than backbone [3, 4, 5] -> neck [3, 4, 5, 6, 7] are not built')
        self.anchors = Anchors(pyramid_levels=self.head_feature_pyramid_levels, logger=logger, **cfg['anchors_config'])
        num_anchors = self.anchors.num_anchors_per_cell     # (the reference hard-codes ratios [1, 2] x 3 scales = 6, :113-116)
        self.head_version = 1
        nlev = len(self.head_feature_pyramid_levels)
        self.regressor = Regressor(fpn_features_num, num_anchors=num_anchors, num_pyramid_levels=nlev, logger=logger,
                                   **cfg['head_config'])
        self.classifier = Classifier(fpn_features_num, num_anchors=num_anchors, num_classes=num_classes,
                                     num_pyramid_levels=nlev, logger=logger, **cfg['head_config'])
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()
        self.focalLoss = FocalLoss()
        self._channels_last_params()
        self.freeze_bn()                                    # ksevendet.py:164
        self._plans = {}

    def _build_backbone(self, cfg, **kwargs):
        """ksevendet.py:166-183."""
        assert cfg['backbone'] in registry._module_to_models, 'Backbone: %s not support.' % cfg['backbone']
        info = {'features_only': True, 'backbone_feature_pyramid_level': self.backbone_feature_pyramid_levels}
        if cfg.get('variant', None):
            assert registry.is_model(cfg['variant']), 'Variant: %s not support.' % cfg['variant']
            return registry.model_entrypoint(cfg['variant'])(**info, **kwargs)
        raise NotImplementedError('a backbone_config without `variant` is not supported (the reference exits here too, '
                                  'ksevendet.py:180-183)')

    # the launch-plan plumbing is the RetinaNet one
    _channels_last_params = _retina.RetinaNet._channels_last_params
    freeze_bn = _retina.RetinaNet.freeze_bn
    _plan = _retina.RetinaNet._plan

    @property
    def conv1(self):                                        # (device check of RetinaNet._plan)
        return self.backbone.conv1

    def forward(self, img_batch, annotations=None, return_head=False, return_loss=False):
        return _retina.RetinaNet.forward(self, img_batch, annotations, return_head, return_loss)

    def set_onnx_convert_info(self, fixed_size=(512, 512)):
        raise NotImplementedError('ONNX export is out of scope (SURVEY.md 2)')
