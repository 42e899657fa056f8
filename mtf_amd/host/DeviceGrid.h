/*
 * DeviceGrid.h -- mtf::hip::Grid: GridTracker<SSM> (SM/include/mtf/SM/GridTracker.h, SM/src/GridTracker.cc:97-392) over ONE batch of
 * grid_size_x * grid_size_y patch trackers on the device.  The reference holds a vector of TrackerBase* and loops over them
 * (update :254-261, resetTrackers :345-392); here the patches are the targets of one mtfhip_batch and a frame is one C-ABI call
 * (mtfhip_grid_frame: every patch's whole update() in one launch for ICLK with a constant Hessian) plus, with
 * reset_at_each_frame = 1, the re-initialisation of the patch trackers on the new grid (mtfhip_grid_reset, not waited for).
 * fb_err_thresh > 0 (the shipped configuration): backwardEstimation (:294-343) is the second batch pass of mtfhip_grid_frame_fb on the
 * previous frame, which the context keeps resident (mtfhip_image_keep_prev: the two frames alternate between two device buffers).
 * TrackerBase-shaped: setImage / initialize / update / setRegion / getRegion with the reference's parameter block.
 *
 * The robust fit of the grid SSM to the patch centroids -- ssm.estimateWarpFromPts (SSM/src/Homography.cc:885-897, Affine.cc:359-369 ->
 * utils::estimateHomography / estimateAffine: RANSAC / LMedS; SURVEY.md section 2: out of scope) -- is a std::function the
 * maintainer points at the SSM's own estimateWarpFromPts; the default is an all-points least-squares fit.
 */
#ifndef MTF_AMD_HOST_DEVICE_GRID_H
#define MTF_AMD_HOST_DEVICE_GRID_H

#include <functional>
#include <vector>

#include "HipModels.h"
#include "SearchMethod.h"

namespace mtf {

#ifdef MTF_AMD_USE_OPENCV
typedef cv::Point2f GridPt;
#else
struct GridPt { float x = 0, y = 0; };   /* cv::Point2f (GridTracker.h:102-103) */
#endif

/* GridTrackerParams (SM/src/GridTracker.cc:20-94); class defaults GridTracker.h:8-24 / Config/parameters.h:505-512 */
struct GridTrackerParams {
	int grid_size_x = 10, grid_size_y = 10;
	int patch_size_x = 10, patch_size_y = 10;
	int reset_at_each_frame = 1;
	bool dyn_patch_size = false;
	bool patch_centroid_inside = true;
	double fb_err_thresh = 0;   /* > 0: forward-backward error estimation (GridTracker.cc:186-190); shipped Config/modules.cfg:81: 2 */
	bool fb_reinit = true;      /* GridTracker.h: GT_FB_REINIT; shipped Config/modules.cfg:82: 1 */
	int n_model_pts = 4;        /* est_params.n_model_pts (SSMEstimatorParams.cc:63; shipped Config/modules.cfg:39) */
	int getResX() const { return resx(); }
	int getResY() const { return resy(); }
	mtfhip_grid_desc desc() const {
		return mtfhip_grid_desc{grid_size_x, grid_size_y, patch_size_x, patch_size_y, reset_at_each_frame, dyn_patch_size ? 1 : 0, patch_centroid_inside ? 1 : 0};
	}
private:
	int resx() const { return grid_size_x + ((dyn_patch_size || patch_centroid_inside) ? 1 : 0); }   /* updateRes :86-94 */
	int resy() const { return grid_size_y + ((dyn_patch_size || patch_centroid_inside) ? 1 : 0); }
};

namespace hip {

class Grid {
public:
	/* estimateWarpFromPts(state_update, mask, in_pts, out_pts, est_params) */
	typedef std::function<void(VectorXd &state_update, const std::vector<GridPt> &in_pts, const std::vector<GridPt> &out_pts)> Estimator;
	std::string name = "grid_hip";

	/* patch trackers: patch_sm (MTFHIP_SM_*) + patch_am + patch_ssm at patch_size x patch_size sampling (mtf.h:782-788: resx = resy =
	 * grid_patch_size), parameters patch_params; grid_ssm: the SSM the grid is laid out with and the estimator's parameterisation */
	Grid(const GridTrackerParams &params, int patch_sm, int patch_am, int patch_ssm, const nt::SMParams &patch_params,
		int grid_ssm = MTFHIP_SSM_HOMOGRAPHY, int device = 0, void *stream = nullptr);
	~Grid();
	Grid(const Grid &) = delete;
	Grid &operator=(const Grid &) = delete;

	void setImage(const ImageView &img);
#ifdef MTF_AMD_USE_OPENCV
	void setImage(const cv::Mat &img) { setImage(imageView(img)); }   /* TrackerBase.h:22 */
#endif
	void initialize(const CornersT &corners);   /* GridTracker.cc:233-246 */
	void update();                              /* :247-285 */
	void setRegion(const CornersT &corners);    /* :287-292 */
	const CornersT &getRegion() { return region; }
	void setEstimator(Estimator e) { estimator = e; }

	const std::vector<GridPt> &getPrevPts() const { return prev_pts; }
	const std::vector<GridPt> &getCurrPts() const { return curr_pts; }
	const VectorXd &getSSMUpdate() const { return ssm_update; }
	const std::vector<int> &getPatchIters() const { return n_iters; }
	/* forward-backward estimation (fb_err_thresh > 0): where every patch tracker came back to on the previous frame and which ones were kept */
	const std::vector<GridPt> &getFbPrevPts() const { return fb_prev_pts; }
	const std::vector<unsigned char> &getFbErrMask() const { return fb_err_mask; }
	/* the corners the last reset handed the patch trackers (n x 8, CornersT layout) and the patch trackers' regions after update() */
	const std::vector<double> &getPatchCorners() const { return patch_corners; }
	const std::vector<double> &getPatchRegions() const { return patch_regions; }
	mtfhip_batch *batch() { return b; }
	const mtfhip_sm_desc &desc() const { return d; }
	const mtfhip_grid_desc &gridDesc() const { return gd; }
private:
	GridTrackerParams params;
	mtfhip_grid_desc gd;
	mtfhip_sm_desc d;
	mtfhip_ctx *ctx = nullptr;
	mtfhip_batch *b = nullptr;
	int n, grid_ssm, n_channels = 1;
	bool reinit_at_each_frame, have_pending = false, have_template = false;
	CornersT region, pending_region;
	std::vector<GridPt> prev_pts, curr_pts, fb_prev_pts;
	std::vector<unsigned char> fb_err_mask;
	bool enable_fb_err_est = false;
	mtfhip_grid_fb_desc fbd{0, 1, 4};
	std::vector<float> cen, fb_cen, prev_f, prev_masked, curr_masked;
	std::vector<int> n_iters;
	std::vector<double> patch_corners, patch_regions;
	VectorXd ssm_update;
	Estimator estimator;
	void resetTrackers(bool reinit);
	static void leastSquaresFit(int ssm, VectorXd &state_update, const std::vector<GridPt> &in_pts, const std::vector<GridPt> &out_pts);
};

} // namespace hip
} // namespace mtf
#endif
