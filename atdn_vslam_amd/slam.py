"""`NeuralSLAM` — the caller-facing state machine of atdn_vslam/slam_framework/neural_slam.py on the MI355X path.

Same states (idle -> odometry -> mapping -> relocalization), same call pattern and the same files on disk
(`<keyframes_path>/rgb/%06d.pth` uint8 frames, `poses.pth` [K,12], `MappingVAE_weights.pth`), with the networks
replaced by the HIP modules of this package:

* odometry mode (`neural_slam.py:192-227`): resize to 376x1232, flow (12 iterations), CLVO head, float32 pose
  accumulation, keyframe decision (10 degrees / 15 m since the last keyframe, `neural_slam.py:268-283`);
* relocalization mode (`neural_slam.py:355-399`): MappingVAE embedding of the query, nearest keyframe embedding,
  flow-based refinement against that keyframe's stored frame.

`end_odometry()` persists the keyframe poses like the reference and then creates the map: with trained weights at hand
(`mapping_weights=` or `<keyframes_path>/MappingVAE_weights.pth`) it loads them, otherwise it trains the MappingVAE on
the keyframes (`mapping.create_map`, the reference's `__create_map`: 50 epochs of AdamW on stock PyTorch, a one-off per
mapping run) — then every keyframe is embedded on the HIP path and the state machine enters relocalization.
"""
import copy
import glob
import math
import os

import torch

from . import transforms
from .modules import ATDNVO, MappingVAE, RAFTGMA
from .pipeline import SLAM_SIZE, resize_frames


class Frame:
    """Keyframe record (slam_framework/frame.py): file of the stored frame, predicted pose, latent embedding."""

    def __init__(self, rgb_file_name, pred_pose, code=None):
        self.rgb_file_name = rgb_file_name
        self.pose = pred_pose
        self.embedding = code


class KeyframePolicy:
    """A frame becomes a keyframe when the motion accumulated since the last one exceeds 10 degrees (norm of the yxz
    Euler vector) or 15 m (`NeuralSLAM.__decide_keyframe`, neural_slam.py:268-283). float32 like the reference."""

    def __init__(self, rot_threshold_deg=10.0, translation_threshold=15.0):
        self.rotation_threshold = (rot_threshold_deg / 180) * math.pi
        self.translation_threshold = translation_threshold
        self.propagation = torch.eye(4, dtype=torch.float32)

    def __call__(self, pred_mat):
        self.propagation = self.propagation @ torch.as_tensor(pred_mat, dtype=torch.float32)
        rotation = transforms.matrix2euler(self.propagation[:3, :3])
        translation = self.propagation[:3, -1]
        if torch.norm(rotation) > self.rotation_threshold or torch.norm(translation) > self.translation_threshold:
            self.propagation = torch.eye(4, dtype=torch.float32)
            return True
        return False


def _load_weights(w):
    return torch.load(w, map_location="cpu") if isinstance(w, (str, os.PathLike)) else w


def _homogeneous(poses12):
    p = torch.as_tensor(poses12, dtype=torch.float32).view(-1, 3, 4)
    last = torch.tensor([0.0, 0.0, 0.0, 1.0]).view(1, 1, 4).repeat(len(p), 1, 1)
    return torch.cat([p, last], dim=1)


class NeuralSLAM:
    """Drop-in for `atdn_vslam.slam_framework.neural_slam.NeuralSLAM`.

    args: object with `.device` and `.keyframes_path` (the reference's `Arguments`).
    odometry_weights / flow_weights / mapping_weights: checkpoint path or state dict. `flow_weights` defaults to the
    path the reference's `GMA_Parameters` names. `map_options`: keyword arguments for `mapping.create_map` when
    `end_odometry()` has to train the map itself.
    """

    FLOW_CHECKPOINT = "atdn_vslam/checkpoints/gma-kitti.pth"  # utils/gma_parameters.py

    def __init__(self, args, odometry_weights=None, start_mode=None, flow_weights=None, mapping_weights=None,
                 precision=None, map_options=None):
        self._args = args
        self._map_options = dict(map_options or {})   # keyword arguments of mapping.create_map (e.g. num_epochs)
        self._base = args.keyframes_path
        self._device = torch.device(args.device if getattr(args, "device", None) not in (None, "cpu") else "cuda:0")
        # frame-by-frame caller that returns a host pose per call (one device synchronisation each): the split-f16 range guard
        # is read after EVERY forward, so no pose computed from clamped activations is ever returned
        self._flow_net = RAFTGMA(max_batch=1, precision=precision, saturation_check_every=1, low_latency=True)
        self._flow_net.load_state_dict(_load_weights(flow_weights if flow_weights is not None else self.FLOW_CHECKPOINT))
        self._flow_net = self._flow_net.to(self._device).eval()
        self._padder = transforms.InputPadder((3,) + SLAM_SIZE)
        self._odometry_net = ATDNVO()
        self._odometry_net.load_state_dict(_load_weights(odometry_weights))
        self._odometry_net = self._odometry_net.to(self._device).eval()
        self._image_buffer = None
        self._mapping_net = None
        self._keyframes = []
        self._policy = KeyframePolicy()
        self._current_pose = torch.eye(4, dtype=torch.float32)

        if start_mode == "mapping":
            self._load_keyframes(embed=False)
            self._mode = "odometry"
            self.end_odometry(mapping_weights)
        elif start_mode == "relocalization":
            w = mapping_weights if mapping_weights is not None else os.path.join(self._base, "MappingVAE_weights.pth")
            self._set_mapping_net(w)
            self._load_keyframes(embed=True)
            self._mode = "relocalization"
        else:
            os.makedirs(os.path.join(self._base, "rgb"), exist_ok=True)
            for f in glob.glob(os.path.join(self._base, "rgb", "*")):
                os.remove(f)
            # a cold start owns the directory: poses and map weights of an earlier session go too
            for stale in ("poses.pth", "MappingVAE_weights.pth"):
                stale_path = os.path.join(self._base, stale)
                if os.path.exists(stale_path):
                    os.remove(stale_path)
            self._mode = "idle"

    # ------------------------------------------------------------------ state machine
    def start_odometry(self):
        if self._mode == "idle":
            self._mode = "odometry"
        else:
            print("Odometry cannot be performed in current SLAM stage")

    def end_odometry(self, mapping_weights=None):
        """Persist the keyframe poses (`poses.pth`, [K,12]), create the map (train the MappingVAE on the keyframes, as
        neural_slam.py:160 does; skipped only when `mapping_weights` is given explicitly), embed every keyframe and
        enter relocalization."""
        if self._mode == "odometry" and len(self._keyframes) > 0:
            poses = torch.stack([kf.pose.flatten()[:12] for kf in self._keyframes], dim=0)
            torch.save(poses, os.path.join(self._base, "poses.pth"))
            self._mode = "mapping"
            default = os.path.join(self._base, "MappingVAE_weights.pth")
            if mapping_weights is None:
                # map creation (neural_slam.py:160,305-352): the reference ALWAYS trains the auto-encoder on the current
                # keyframes here and overwrites MappingVAE_weights.pth — a file left in the directory by an earlier
                # session belongs to another environment and is never picked up. Training runs on stock PyTorch, as in
                # the reference; the embedding it yields runs on the HIP path again. Only an explicit `mapping_weights`
                # argument skips the training.
                from .mapping import create_map
                create_map(self._base, device=self._device, **self._map_options)
                mapping_weights = default
            self._set_mapping_net(mapping_weights)
            for kf in self._keyframes:
                kf.embedding = self._embed(torch.load(kf.rgb_file_name))
            self._mode = "relocalization"
        elif len(self._keyframes) == 0:
            print("There is no explored enviromnent yet!")
        elif self._mode == "mapping" and mapping_weights is not None:
            self._set_mapping_net(mapping_weights)
            for kf in self._keyframes:
                kf.embedding = self._embed(torch.load(kf.rgb_file_name))
            self._mode = "relocalization"
        else:
            print("Current state is not odometry")

    @torch.no_grad()
    def __call__(self, im):
        if self._mode == "odometry":
            im = im.to(self._device)
            im = resize_frames(im if im.dtype == torch.uint8 else im.float(), SLAM_SIZE)   # (uint8: converted inside the resize kernel)
            if im.dtype != torch.float32:
                im = im.float()
            if self._image_buffer is not None:
                im2 = self._padder.pad(im)[0]
                # (pair mode's bits, one feature-network pass per frame while the chain of odometry calls is unbroken)
                _, flow = self._flow_net.forward_consecutive(self._image_buffer, im2, iters=12)
                pred_rot, pred_tr = self._odometry_net(flow)
                rot, tr = pred_rot.squeeze().cpu(), pred_tr.squeeze().cpu()
                pred_mat = transforms.transform(rot, tr)
                self._current_pose = transforms.accumulate(self._current_pose, rot, tr)  # float32 pose @ pred_mat
                if self._policy(pred_mat):
                    name = os.path.join(self._base, "rgb", "%06d.pth" % len(self._keyframes))
                    torch.save(im2.to("cpu").byte(), name)
                    self._keyframes.append(Frame(name, self._current_pose))
                self._image_buffer = im2
            else:
                self._image_buffer = self._padder.pad(im)[0]
                name = os.path.join(self._base, "rgb", "000000.pth")
                torch.save(im.to("cpu").byte(), name)
                self._keyframes.append(Frame(name, self._current_pose))
            return self._current_pose
        if self._mode == "relocalization":
            q = im.to(self._device).float()
            if q.dim() == 3:
                q = q.unsqueeze(0)
            return self._relocalize(q)
        raise Exception("SLAM called in invalid state!")

    def mode(self):
        return copy.deepcopy(self._mode)

    def to(self, device):
        self._args.device = device
        self._device = torch.device(device)
        self._flow_net = self._flow_net.to(device)
        self._odometry_net = self._odometry_net.to(device)
        if self._mapping_net is not None:
            self._mapping_net = self._mapping_net.to(device)

    def get_keyframe(self, index):
        return self._keyframes[index]

    def __getitem__(self, index):
        return self._keyframes[index]

    def __len__(self):
        return len(self._keyframes)

    # ------------------------------------------------------------------ internals
    def _set_mapping_net(self, weights):
        self._mapping_net = MappingVAE()
        self._mapping_net.load_state_dict(_load_weights(weights))
        self._mapping_net = self._mapping_net.to(self._device).eval()

    def _embed(self, rgb):
        rgb = rgb.to(self._device).float()
        if rgb.dim() == 3:
            rgb = rgb.unsqueeze(0)
        return self._mapping_net(rgb)[0]

    def _load_keyframes(self, embed):
        poses = _homogeneous(torch.load(os.path.join(self._base, "poses.pth")))
        files = sorted(glob.glob(os.path.join(self._base, "rgb", "*")))
        for i, f in enumerate(files):
            code = self._embed(torch.load(f)) if embed else None
            self._keyframes.append(Frame(f, poses[i], code))

    def _relocalize(self, image):
        mu = self._mapping_net(image)[0]
        distances = torch.stack([torch.norm(kf.embedding - mu, p=2) for kf in self._keyframes], dim=0)
        closest = self._keyframes[int(torch.argmin(distances))]
        initial_pose = closest.pose
        # refinement: odometry between the stored keyframe image and the query (neural_slam.py:386-399)
        im1 = torch.load(closest.rgb_file_name).unsqueeze(0).to(self._device).float()
        _, flow = self._flow_net(im1, image, iters=12, test_mode=True)
        pred_rot, pred_tr = self._odometry_net(flow)
        pose_diff = transforms.transform(pred_rot.squeeze().cpu(), pred_tr.squeeze().cpu())
        return initial_pose, initial_pose @ pose_diff, distances.cpu()
