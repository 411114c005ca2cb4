// OptDenseHIP.hpp -- the reference's option table (include/optdense.h: namespace OPTDENSE of libs/MVS/DepthMap.cpp:67-114, loaded from `--dense-config-file`)
// as the options of the resident-scene driver (include/DenseDepthMapsHIP.hpp).  Header-only glue; link libmvsfront.so for the table itself.
//
//   MVSFOptDense od;  mvsf_optdense_load("dense.ini", &od, nullptr);   // init() + Load + update(): defaults if there is no file
//   od.nNumViews = 8;                                                   // command-line flags are assigned after update() (DensifyPointCloud.cpp:241-252)
//   MVS::DenseDepthMapsHIP::Options opt = MVS::DenseOptionsFrom(od, /*seed*/ 1);
//   MVSFOptions front;  mvsf_optdense_front(&od, &front);               // view selection and depth initialisation (include/mvsfront.h)
#pragma once
#include "optdense.h"
#include "DenseDepthMapsHIP.hpp"

namespace MVS {

inline DenseDepthMapsHIP::Options DenseOptionsFrom(const MVSFOptDense& o, uint32_t seed = 0) {
	DenseDepthMapsHIP::Options d;
	mvsf_optdense_estimator(&o, &d);                       // the PMHipParams part
	d.seed = seed;
	d.nOptimize = o.nOptimize;                             // DepthFlags, SceneDensify.cpp:1884-1886,1919-1920,1955
	d.nSpeckleSize = o.nSpeckleSize; d.nIpolGapSize = o.nIpolGapSize;
	d.fDepthDiffThreshold = o.fDepthDiffThreshold; d.fNormalDiffThreshold = o.fNormalDiffThreshold;
	d.nMinViewsFilter = o.nMinViewsFilter; d.nMinViewsFilterAdjust = o.nMinViewsFilterAdjust; d.nMinViewsFuse = o.nMinViewsFuse;
	d.bFilterAdjust = o.bFilterAdjust != 0;
	d.bEstimateColor = o.nEstimateColors == 2;             // FuseDepthMaps(pointcloud, nEstimateColors == 2, nEstimateNormals == 2), SceneDensify.cpp:1697-1700
	d.bEstimateNormal = o.nEstimateNormals == 2;
	return d;
}

} // namespace MVS
