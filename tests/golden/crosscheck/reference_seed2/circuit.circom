
template Program() {
   var yOGWiyKq = (4755818798689304646>=297835709732521026);
   var bMovmdJM = 1153784141284902224;
   for (var Purecawx = 24; (Purecawx<=48); Purecawx=(Purecawx+1)) {
      bMovmdJM=(4643249767881755238<=7839643633482800064);
      signal output tcRgYCbE[9];
   }
   for (var ssZoRCls = 19; (ssZoRCls<=40); ssZoRCls=(ssZoRCls+1)) {
      ssZoRCls=(219322712127010475^7345385423574670337);
   }
   for (var VifsBwYD = 20; (VifsBwYD<=32); VifsBwYD=(VifsBwYD+1)) {
      signal input NOpAuZlw[10];
      bMovmdJM=(15114573148747454923/5260112035621119752);
   }
   yOGWiyKq=(3122423844521071056||17863132340689033660);
   for (var DjkyHyWa = 27; (DjkyHyWa<=46); DjkyHyWa=(DjkyHyWa+1)) {
      for (var MBmqZUte = 2; (MBmqZUte<=9); MBmqZUte=(MBmqZUte+1)) {
         MBmqZUte=(8220876534833942642<2768873974381355514);
         bMovmdJM=18087705744095899905;
         DjkyHyWa=(820654990933447174>>4609372707992676344);
         signal output GLvKIEUq[4];
      }
      DjkyHyWa=(14689601165245740032&5334209129747575313);
      var uZzRHOBr = (2187804329627890100|8233126757333406315);
   }
   yOGWiyKq=(16112811622061952854<=10502255204065941319);
   for (var EReeRrSA = 1; (EReeRrSA<=34); EReeRrSA=(EReeRrSA+1)) {
      var AbrzxzfO[8];
      yOGWiyKq=(11495610054904088730>1197274351933821800);
      AbrzxzfO[6]=(7566059052750409092**8073400715132996404);
   }
   for (var xcqATPSZ = 26; (xcqATPSZ<=34); xcqATPSZ=(xcqATPSZ+1)) {
      signal output iZbzbRdd[10];
      for (var YpuPEobK = 13; (YpuPEobK<=36); YpuPEobK=(YpuPEobK+1)) {
         yOGWiyKq=(16471258040983990514|8148584215139850159);
      }
   }
   yOGWiyKq=(11610019366687507168>=2936696593097521062);
   var noTdpvzm = (7668244031874758893&17001549348262936080);
   for (var VQGOWNDJ = 1; (VQGOWNDJ<=23); VQGOWNDJ=(VQGOWNDJ+1)) {
      bMovmdJM=(11427248539843374419+8812512598567838525);
      for (var EOyEtlQM = 22; (EOyEtlQM<=25); EOyEtlQM=(EOyEtlQM+1)) {
         for (var YrUAWXhW = 15; (YrUAWXhW<=45); YrUAWXhW=(YrUAWXhW+1)) {
            EOyEtlQM=(15831721695335336843>=3605422612146781076);
            for (var LXQXcrRS = 25; (LXQXcrRS<=57); LXQXcrRS=(LXQXcrRS+1)) {
               var uJzXrCNo = (11814345791230811641&&3091386535241765342);
               for (var KKblSJnU = 28; (KKblSJnU<=33); KKblSJnU=(KKblSJnU+1)) {
                  YrUAWXhW=(9645350584152184360|1058605219832512629);
               }
               signal output ZzmtISyQ[10];
            }
         }
      }
   }
   for (var xOZxHBnk = 4; (xOZxHBnk<=49); xOZxHBnk=(xOZxHBnk+1)) {
      xOZxHBnk=(14225794060085926829&&4651087882203696563);
      for (var ZUYzQFFD = 2; (ZUYzQFFD<=49); ZUYzQFFD=(ZUYzQFFD+1)) {
         for (var BnKYroSC = 18; (BnKYroSC<=31); BnKYroSC=(BnKYroSC+1)) {
            for (var FbIgCDAa = 15; (FbIgCDAa<=23); FbIgCDAa=(FbIgCDAa+1)) {
               var zgHWPdPS = (12610590513044013138%1012460998714680484);
            }
         }
         for (var FPmsigxx = 15; (FPmsigxx<=59); FPmsigxx=(FPmsigxx+1)) {
            var YxYtpsvZ[2];
            signal output rFUiHpgR[5];
         }
         noTdpvzm=(2767353162970855811%8839953866124815724);
      }
   }
   for (var qxOTdGcS = 29; (qxOTdGcS<=48); qxOTdGcS=(qxOTdGcS+1)) {
      for (var yZNAVkJh = 14; (yZNAVkJh<=27); yZNAVkJh=(yZNAVkJh+1)) {
         noTdpvzm=(11304989669753675203&8131756980029101967);
      }
      for (var fsZfcTDO = 27; (fsZfcTDO<=37); fsZfcTDO=(fsZfcTDO+1)) {
         var poWEGDcO[5];
         yOGWiyKq=(920375292935234407*4363060293246474472);
         noTdpvzm=(5948475023341551741|3051230921905402045);
      }
      var hVyoeABm = (13105899489838646807+10390354983974515345);
   }
   bMovmdJM=(4994154185491118064^8384755663227392338);
   noTdpvzm=(3167498534171094769<<12384501845717211637);
}

component main = Program();
        