
template Program() {
   signal output yFZxaLSp;
   yFZxaLSp<==yFZxaLSp;
   var sNDGhzgj[2];
   for (var APRbyuoW = 24; (APRbyuoW<=49); APRbyuoW=(APRbyuoW+1)) {
      APRbyuoW=(13572523122062659791>15708454227269489642);
      APRbyuoW=(3397266985847803599^11697243335617396588);
   }
   for (var qZjwapZu = 15; (qZjwapZu<=42); qZjwapZu=(qZjwapZu+1)) {
      var CpKwNtnU = (1893784285118007885*17575622708178026303);
      sNDGhzgj[0]=(6038542295364746031-1675065490048566956);
   }
   var lIvSjpKc = (1900725450149394473^10192793590960229490);
   for (var ebTLqBEB = 25; (ebTLqBEB<=27); ebTLqBEB=(ebTLqBEB+1)) {
      sNDGhzgj[1]=(3839522382823317159>=16530416909290103009);
      signal output JmBkkSEJ;
      sNDGhzgj[1]=(1361588921065071273>=3293483189413264795);
      for (var hFAkmNcz = 20; (hFAkmNcz<=60); hFAkmNcz=(hFAkmNcz+1)) {
         JmBkkSEJ<==(6844699814287011198>2838839461188090824);
         lIvSjpKc=(6422020686336550364>=10425774258633789735);
      }
   }
   signal output XoiDtHpw[5];
   sNDGhzgj[0]=(6986238359604814027/11072984369085759742);
   signal output pRyizSdg[9];
   sNDGhzgj[0]=(3546601466037530926/4382596931893952953);
   var YfpRHBZN = (3578896991783952467==15413836824559724329);
   for (var HdhgwlEk = 30; (HdhgwlEk<=33); HdhgwlEk=(HdhgwlEk+1)) {
      HdhgwlEk=(sNDGhzgj[1]);
   }
   sNDGhzgj[1]=(8553271723344167844|14357742889006232375);
   YfpRHBZN=(8401976234875110884%6549343263556537402);
   YfpRHBZN=(12959474280816881923||10912299899699700192);
   for (var ppfLJbkk = 14; (ppfLJbkk<=29); ppfLJbkk=(ppfLJbkk+1)) {
      ppfLJbkk=(13942511405078590497%5398936254385910566);
      lIvSjpKc=(12783492559782974005>7652246671878467661);
      sNDGhzgj[1]=(sNDGhzgj[0]);
      signal input tuXkIXbZ[3];
   }
   for (var OZzyKvqH = 1; (OZzyKvqH<=38); OZzyKvqH=(OZzyKvqH+1)) {
      signal output yYtqDooX[8];
      var aXnDnMxx = (4230106202331957872-209331348739352257);
      sNDGhzgj[1]=(1460904038040973336+16674739611096241207);
   }
   lIvSjpKc=(1254748722376948257||11588117145265977936);
   var ymiLFXoW[9];
   for (var HLMuHKQs = 21; (HLMuHKQs<=37); HLMuHKQs=(HLMuHKQs+1)) {
      HLMuHKQs=(17352293099979198377&&13392486730783246351);
   }
}

component main = Program();
        