"""train_vae — VAE pre-training driver with the reference's CLI (vae/train_vae.py:47-161), GPU-resident and data-parallel.

Same flags, dataset layout (<dataset>/rgb/*.png, <dataset>/segmentation/*.png), preprocessing, 10 % validation split,
model naming, restart prompt and early-stopping loop as the reference.  What changed underneath: the frame tables are
uploaded to HBM once, every SGD step runs in HIP kernels, and under torch.distributed.run each rank trains its slice of
every global minibatch with an RCCL gradient all-reduce (rank 0 alone prints, prompts and saves).

    python train_vae.py --dataset data --z_dim 64
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_vae.py --batch_size 4096
"""
import argparse
import os
import shutil
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))                       # makes `mi355` importable when cwd = vae/ (reference layout)

from vae.models import ConvVAE, MlpVAE, bce_loss, bce_loss_v2, mse_loss  # noqa: E402


def preprocess_rgb_frame(frame):
    return frame[:, :, :3].astype(np.float32) / 255.0                # RGBA -> RGB in [0,1]   (vae/train_vae.py:15-18)


def preprocess_seg_frame_road_only(frame):
    return (frame[:, :, :1] == 7).astype(np.float32)                 # binary road mask        (:20-24)


def preprocess_seg_frame(frame):
    return frame[:, :, :1].astype(np.float32) / 12.0                 # class id 0..12 -> [0,1] (:26-29)


def rgb_frame_u8(frame):
    return np.ascontiguousarray(frame[:, :, :3])                     # RGBA -> RGB, still uint8: /255 happens on the device (VAE._frames)


def load_images(dir_path, preprocess_fn, workers=None):
    """Same result as the reference loader (vae/train_vae.py:31-39): every *.png of dir_path in os.listdir order, preprocessed and stacked.
    Decoding runs on a thread pool (PIL releases the GIL while it inflates); order is that of the directory listing, not of completion."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    names = [f for f in os.listdir(dir_path) if os.path.splitext(f)[1] == ".png"]
    if not names:
        raise FileNotFoundError("no .png frames in %s" % dir_path)

    def one(name):
        with Image.open(os.path.join(dir_path, name)) as im:
            return preprocess_fn(np.asarray(im))
    workers = workers if workers is not None else min(32, os.cpu_count() or 1)
    if workers <= 1 or len(names) < 4:
        return np.stack([one(n) for n in names], axis=0)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return np.stack(list(pool.map(one, names)), axis=0)


def train_val_split(images, val_portion=0.1):
    val_split = int(images.shape[0] * val_portion)
    return images[val_split:], images[:val_split]                    # first 10 % = validation      (:41-45)


def main(argv=None):
    parser = argparse.ArgumentParser(description="Trains a VAE with RGB images as source and RGB or segmentation images as target")
    parser.add_argument("--model_name", type=str, default=None)
    parser.add_argument("--dataset", type=str, default="data")
    parser.add_argument("--use_segmentation_as_target", type=bool, default=False)
    parser.add_argument("--loss_type", type=str, default="bce")
    parser.add_argument("--model_type", type=str, default="cnn")
    parser.add_argument("--beta", type=int, default=1)
    parser.add_argument("--z_dim", type=int, default=64)
    parser.add_argument("--learning_rate", type=float, default=1e-4)
    parser.add_argument("--lr_decay", type=float, default=1.0)
    parser.add_argument("--batch_size", type=int, default=100)
    parser.add_argument("--kl_tolerance", type=float, default=0.0)
    parser.add_argument("-restart", action="store_true")
    parser.add_argument("--max_epochs", type=int, default=0, help="(new) stop after this many epochs; 0 = early stopping only")
    parser.add_argument("--host_float_frames", action="store_true", help="(new) convert the RGB frames to float32 / 255 on the host like the reference")
    args = parser.parse_args(argv)

    from mi355 import dist as midist
    world, rank, _ = midist.init_from_env()
    chief = rank == 0

    # RGB frames stay uint8 on the host (4x less memory and upload); the model normalises them on the device with the same float32 division
    rgb_images = load_images(os.path.join(args.dataset, "rgb"), preprocess_rgb_frame if args.host_float_frames else rgb_frame_u8)
    seg_images = load_images(os.path.join(args.dataset, "segmentation"), preprocess_seg_frame) if args.use_segmentation_as_target else None

    np.random.seed(0)                                                # every rank: identical permutations
    train_source_images, val_source_images = train_val_split(rgb_images, val_portion=0.1)
    if args.use_segmentation_as_target:
        train_target_images, val_target_images = train_val_split(seg_images, val_portion=0.1)
    else:
        train_target_images, val_target_images = train_source_images, val_source_images
    source_shape = train_source_images.shape[1:]
    target_shape = train_target_images.shape[1:] if args.use_segmentation_as_target else source_shape

    if args.model_name is None:
        args.model_name = "{}_{}_{}_zdim{}_beta{}_kl_tolerance{}_{}".format(
            "seg" if args.use_segmentation_as_target else "rgb", args.loss_type, args.model_type, args.z_dim, args.beta,
            args.kl_tolerance, os.path.splitext(os.path.basename(args.dataset))[0])
    if chief:
        print("train_source_images.shape", train_source_images.shape)
        print("val_source_images.shape", val_source_images.shape)
        print("Training parameters:")
        for k, v in vars(args).items():
            print(f"  {k}: {v}")

    losses = {"bce": bce_loss, "bce_v2": bce_loss_v2, "mse": mse_loss}
    if args.loss_type not in losses:
        raise Exception("No loss function \"{}\"".format(args.loss_type))
    if args.model_type not in ("cnn", "mlp"):
        raise Exception("No model type \"{}\"".format(args.model_type))
    VAEClass = ConvVAE if args.model_type == "cnn" else MlpVAE
    vae = VAEClass(source_shape=source_shape, target_shape=target_shape, z_dim=args.z_dim, beta=args.beta,
                   learning_rate=args.learning_rate, lr_decay=args.lr_decay, kl_tolerance=args.kl_tolerance,
                   loss_fn=losses[args.loss_type], model_dir=os.path.join("models", args.model_name))

    if not args.restart and chief and world == 1:
        if os.path.isdir(vae.log_dir) and len(os.listdir(vae.log_dir)) > 0:
            answer = input("Model \"{}\" already exists. Do you wish to continue (C) or restart training (R)? ".format(args.model_name))
            if answer.upper() == "R":
                args.restart = True
            elif answer.upper() != "C":
                raise Exception("There are already log files for model \"{}\". Please delete it or change model_name and try again".format(args.model_name))
    if args.restart and chief:
        shutil.rmtree(vae.model_dir)
        for d in vae.dirs:
            os.makedirs(d)
    midist.barrier()
    vae.init_session(init_logging=chief)
    if not args.restart:
        vae.load_latest_checkpoint()

    min_val_loss, counter = float("inf"), 0
    if chief:
        print("Training")
    while True:
        epoch = vae.get_step_idx()
        if chief and (epoch + 1) % 10 == 0:
            print(f"Epoch {epoch + 1}")
        val_loss, _ = vae.evaluate(val_source_images, val_target_images, args.batch_size)
        if val_loss < min_val_loss:                                  # early stopping on the validation reconstruction loss
            counter, min_val_loss = 0, val_loss
            vae.save()
        else:
            counter += 1
            if counter >= 10:
                if chief:
                    print("No improvement in last 10 epochs, stopping")
                break
        if args.max_epochs and epoch >= args.max_epochs:
            break
        vae.train_one_epoch(train_source_images, train_target_images, args.batch_size)


if __name__ == "__main__":
    main()
