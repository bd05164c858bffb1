# copies the summaries a refresh_profiles.sh run left under gpurun_out/profiles_<round>/ into profiles/ (tracked)
RND=${1:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/gpurun_out/profiles_$RND/* $R/profiles/ && ls $R/profiles
