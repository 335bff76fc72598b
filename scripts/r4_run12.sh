bash scripts/ab.sh "WDM_X=0" "WDM_TMP_SKIP_FIN=1" "WDM_X=0" "WDM_TMP_SKIP_FIN=1"
