"""The call sequence of the reference's pose_optimization.py (PoseOptimizer.__init__ :99-175 and
optimize_poses :177-240), restated so that it can run where /root/reference is not mounted (the GPU box).
It only uses names the reference imports from `lib_python`."""
import os

CV_8UC1, CV_32FC3 = 0, 21  # the two OpenCV constants pose_optimization.py imports from cv2


def build_pose_optimizer(lib, base_dir, model_type, frames, opt_params, use_global_scale=False):
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base_dir, False)
    dv.createColorStream("full", "color_full", ".png", CV_32FC3)
    dv.createColorStream("down", "color_down", ".raw", CV_32FC3)
    if os.path.isdir(os.path.join(base_dir, "dynamic_mask")):
        dv.createColorStream("dynamic_mask", "dynamic_mask", ".png", CV_8UC1)
    depth_tag = f"depth_{model_type}"
    dv.createDepthStream(depth_tag, depth_tag, [-1, -1])
    dv.printInfo()
    dv.save()
    fcp = lib.FlowConstraintsParams()
    fcp.frameRange.resolve(dv.numFrames(), True)
    fc = lib.FlowConstraintsCollection(dv, fcp)
    fc.setStaticFlagFromDynamicMask(8)
    fc.save()
    return dv, fc


def optimize_poses(lib, dv, fc, frames, opt_params, use_global_scale=False):
    frames_string = ",".join(str(x) for x in frames)
    dv.clearDepthCaches()
    processor = lib.DepthVideoProcessor(dv)
    params = lib.DepthVideoProcessor.Params()
    params.depthStream = dv.numDepthStreams() - 1
    params.frameRange.fromString(frames_string)
    params.poseOptimizer = opt_params
    params.poseOptimizer.frameRange.fromString(frames_string)

    params.op = lib.DepthVideoProcessor.Op.ResetDepthXforms
    params.depthXformDesc.type = lib.XformType.Depth
    params.depthXformDesc.depthType = lib.DepthXformType.Global
    params.depthXformDesc.valueXform = lib.ValueXformType.Scale
    processor.process(params)

    params.op = lib.DepthVideoProcessor.Op.ResetSpatialXforms
    params.spatialXformDesc.type = lib.XformType.Spatial
    params.spatialXformDesc.spatialType = lib.SpatialXformType.Identity
    params.spatialXformDesc.valueXform = lib.ValueXformType.Scale
    processor.process(params)

    processor.normalizeDepth(params, fc)
    processor.optimizePoses(params, fc)

    if use_global_scale:
        params.poseOptimizer.fixPoses = True
        params.poseOptimizer.numSteps = 1
        params.poseOptimizer.coarseToFine = False
        params.op = lib.DepthVideoProcessor.Op.ResetDepthXforms
        processor.process(params)
        params.op = lib.DepthVideoProcessor.Op.ResetSpatialXforms
        processor.process(params)
        processor.normalizeDepth(params, fc)
        processor.optimizePoses(params, fc)
    dv.save()
    return processor
