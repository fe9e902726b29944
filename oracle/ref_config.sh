#!/bin/sh
# oracle/ref_config.sh <outdir> <refdir> -- TEST INFRASTRUCTURE.
# Writes the two headers the reference's cmake would generate (gtsam/config.h.in:31-90,
# cmake/dllexport.h.in) with the reference's DEFAULT option values, minus TBB/Boost/metis
# which are not available in this image.  Hand-stated; nothing is copied from the reference.
set -e
out="$1"; ref="$2"
mkdir -p "$out"
cat > "$out/config.h" <<EOF
#pragma once
#define GTSAM_VERSION_MAJOR 4
#define GTSAM_VERSION_MINOR 3
#define GTSAM_VERSION_PATCH 0
#define GTSAM_VERSION_NUMERIC 40300
#define GTSAM_VERSION_STRING "4.3a0"
#define GTSAM_SOURCE_TREE_DATASET_DIR "$ref/examples/Data"
#define GTSAM_INSTALLED_DATASET_DIR "$ref/examples/Data"
#define GTSAM_POSE3_EXPMAP
#define GTSAM_ROT3_EXPMAP
#define GTSAM_DT_MERGING
#define GTSAM_EIGEN_VERSION_WORLD 3
#define GTSAM_EIGEN_VERSION_MAJOR 4
#define GTSAM_EIGEN_VERSION_MINOR 0
#define GTSAM_ALLOCATOR_STL
#define GTSAM_THROW_CHEIRALITY_EXCEPTION
#define GTSAM_ALLOW_DEPRECATED_SINCE_V43
#define GTSAM_TANGENT_PREINTEGRATION
EOF
cat > "$out/dllexport.h" <<EOF
#pragma once
#define GTSAM_EXPORT
#define GTSAM_EXTERN_EXPORT extern
EOF
