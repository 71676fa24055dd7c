from .preprocess import batch_convert_for_resnet, load_image, preprocess_batch, resize_bilinear

__all__ = ["load_image", "preprocess_batch", "batch_convert_for_resnet", "resize_bilinear"]
