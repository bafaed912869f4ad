set -e
for d in 0 1 2 4 7; do
  echo "PG_ENC_DBG=$d"
  PG_ENC_DBG=$d bash tools/kt_quick.sh | grep "enc_bwd_persist" | cut -c1-160
done
